#!/bin/bash
O=gpurun_out/r03_exp15; mkdir -p $O
for sh in "128 11 3 16 65536" "128 3 1 16 65536" "64 11 3 16 131072" "64 7 5 16 131072" "256 11 5 16 8192"; do for v in xb0 xb4 xb8 xb0 xb4 xb8; do echo -n "$v $sh  "; timeout 120 tools/kbench/bin/kbench_$v $sh 9 2>&1 | grep -a "time:" | cut -c1-60; done; done > $O/kbench.txt 2>&1; cat $O/kbench.txt
