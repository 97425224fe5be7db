#!/bin/bash
O=gpurun_out/r03_exp13; mkdir -p $O
for sh in "64 11 3 16 131072" "64 7 5 16 131072" "64 3 1 16 131072"; do for v in rawres c64n256 c64n256raw; do echo "== $v $sh"; timeout 120 tools/kbench/bin/kbench_$v $sh 7 2>&1 | grep -a "time:\|check"; KB_ACC=1 timeout 120 tools/kbench/bin/kbench_$v $sh 7 2>&1 | grep -a "time:"; done; done > $O/kbench.txt 2>&1; cat $O/kbench.txt | cut -c1-110
