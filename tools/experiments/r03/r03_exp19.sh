#!/bin/bash
O=gpurun_out/r03_exp19; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_longform.py tests/test_cabi.py -m gpu -q -x --timeout 600 2>&1 | tail -2
bash tools/ab_bench.sh 3 libvtts_prev.so libvtts_hifigan.so > $O/ab.txt 2>&1; cat $O/ab.txt
