#!/bin/bash
O=gpurun_out/r03_exp4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 600 -k "fused_pair_kat" 2>&1 | tail -40 > $O/pytest.txt; grep -a "assert\|Error\|passed\|failed" $O/pytest.txt | head -20
for v in cur cur_res0; do for sh in "32 3 1 2 1100" "32 3 1 16 65536" "32 7 3 2 1100" "32 11 5 2 1100" "64 3 1 2 1100"; do echo "== $v $sh"; timeout 60 tools/kbench/bin/kbench_$v $sh 3 2>&1 | grep -a "check\|time"; done; done
