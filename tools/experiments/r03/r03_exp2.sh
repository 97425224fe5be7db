#!/bin/bash
# round 3, GPU call 2: per-instruction co-issue table, GPU tests of the new build, interleaved A/B of the library variants
O=gpurun_out/r03_exp2; mkdir -p $O
timeout 300 tools/kbench/bin/coissue 400 1 > $O/coissue_fine.txt 2>&1; echo "coissue rc=$?"; cut -c1-400 $O/coissue_fine.txt | tail -25
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
bash tools/ab_bench.sh 3 libvtts_base.so libvtts_blk.so libvtts_res.so libvtts_hifigan.so > $O/ab.txt 2>&1; cat $O/ab.txt
