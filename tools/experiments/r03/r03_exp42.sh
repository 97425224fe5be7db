#!/bin/bash
# with "zigzag" on: do smaller micro-batches (intermediate tensors that fit the 256 MB Infinity Cache) pay now?
O=gpurun_out/r03_exp42; mkdir -p $O
for r in 1 2; do for cfg in "0 1" "32 1" "16 1" "8 1" "32 2" "16 2" "8 2"; do set -- $cfg; echo -n "microbatch $1 streams $2  "; timeout 300 python bench.py --microbatch $1 --streams $2 --steps 6 --warmup 2 --no-cpu-baseline --no-rtf --no-f32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.4e ms %.2f'%(d['value'],d['ms_per_step']))"; done; done > $O/mb.txt 2>&1; cat $O/mb.txt
