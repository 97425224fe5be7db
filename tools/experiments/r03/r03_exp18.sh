#!/bin/bash
O=gpurun_out/r03_exp18; mkdir -p $O
bash tools/ab_bench.sh 3 libvtts_div.so libvtts_hifigan.so > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_longform.py -m gpu -q -x --timeout 600 2>&1 | tail -2
