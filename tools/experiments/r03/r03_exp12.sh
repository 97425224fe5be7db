#!/bin/bash
O=gpurun_out/r03_exp12; mkdir -p $O; R=$PWD
for sh in "32 11 3 16 262144" "32 11 5 16 262144" "32 7 5 16 262144" "32 3 1 16 262144" "32 11 5 2 1100" "32 3 1 2 700" "32 7 3 3 333"; do for v in norawres rawres; do echo "== $v $sh"; timeout 120 tools/kbench/bin/kbench_$v $sh 7 2>&1 | grep -a "time:\|check"; KB_ACC=1 timeout 120 tools/kbench/bin/kbench_$v $sh 7 2>&1 | grep -a "time:\|check"; done; done > $O/kbench.txt 2>&1; grep -a "==\|time\|check" $O/kbench.txt | paste - - - - - | cut -c1-60,75-100,150-200,230-260,320-360
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
bash tools/ab_bench.sh 3 libvtts_norawres.so libvtts_hifigan.so > $O/ab.txt 2>&1; cat $O/ab.txt
