#!/bin/bash
O=gpurun_out/r03_exp5; mkdir -p $O
for v in cur cur_res1; do for sh in "32 3 1 2 1100" "32 3 1 16 65536" "32 7 3 2 1100" "32 11 5 2 1100" "32 11 5 16 65536" "64 3 1 2 1100" "64 11 3 16 32768" "128 7 5 16 16384" "256 11 5 16 4096"; do echo "== $v $sh"; timeout 60 tools/kbench/bin/kbench_$v $sh 3 2>&1 | grep -a "check"; KB_ACC=1 timeout 60 tools/kbench/bin/kbench_$v $sh 3 2>&1 | grep -a "check"; done; done > $O/kbench_check.txt 2>&1; grep -c OK $O/kbench_check.txt; grep -B1 MISMATCH $O/kbench_check.txt
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
bash tools/ab_bench.sh 3 libvtts_pk.so libvtts_hifigan.so libvtts_res.so > $O/ab.txt 2>&1; cat $O/ab.txt
