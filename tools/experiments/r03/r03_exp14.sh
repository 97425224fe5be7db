#!/bin/bash
O=gpurun_out/r03_exp14; mkdir -p $O
for r in 1 2 3; do for c in -1 2; do echo -n "chains $c  "; timeout 300 python bench.py --chains $c --steps 6 --warmup 2 --no-cpu-baseline --no-f32 --no-rtf 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.4e ms %.2f roof %.3f'%(d['value'],d['ms_per_step'],d['roofline']['frac']))"; done; done > $O/ab_chains.txt 2>&1; cat $O/ab_chains.txt
