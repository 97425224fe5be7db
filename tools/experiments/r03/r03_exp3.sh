#!/bin/bash
# round 3, GPU call 3: LDS-DMA / throughput-pattern rows of the co-issue table; packed vs plain f32 multiplies in the whole forward
O=gpurun_out/r03_exp3; mkdir -p $O
timeout 300 tools/kbench/bin/coissue 400 1 > $O/coissue_fine.txt 2>&1; echo "coissue rc=$?"; cut -c1-500 $O/coissue_fine.txt | tail -6
bash tools/ab_bench.sh 3 libvtts_pk.so libvtts_nopk.so libvtts_hifigan.so > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x --timeout 600 2>&1 | tail -2
