#!/bin/bash
O=gpurun_out/r03_exp30; mkdir -p $O
for mb in 0 32 16 8 4; do for st in 1 2; do echo -n "microbatch $mb streams $st  "; timeout 300 python bench.py --microbatch $mb --streams $st --steps 6 --warmup 2 --no-cpu-baseline --no-rtf --no-f32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.4e ms %.2f roof %.3f'%(d['value'],d['ms_per_step'],d['roofline']['frac']))"; done; done > $O/mb.txt 2>&1; cat $O/mb.txt
