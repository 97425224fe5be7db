#!/bin/bash
O=gpurun_out/r03_exp8; mkdir -p $O
for sh in "32 11 3 16 262144" "32 7 3 16 262144" "64 11 3 16 131072" "64 7 3 16 131072" "128 3 1 16 65536" "128 11 3 16 65536"; do echo "== $sh"; timeout 120 tools/kbench/bin/kbench_tl $sh 3 20 2>&1 | grep -a -v "^  *[0-9]* s[0-9]" | head -40; done > $O/timeline.txt 2>&1
cat $O/timeline.txt | grep -a "==\|time:\|stage_x\|issue\|wait\|lrelu\|init_acc\|main loop\|epilogue\|add\|pack\|total\|busy"
