#!/bin/bash
O=gpurun_out/r03_exp31; mkdir -p $O
for r in 1 2 3; do for cfg in "0 1" "32 2" "16 2" "16 4" "22 3" "16 3"; do set -- $cfg; echo -n "microbatch $1 streams $2  "; timeout 300 python bench.py --microbatch $1 --streams $2 --steps 6 --warmup 2 --no-cpu-baseline --no-rtf --no-f32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.4e ms %.2f'%(d['value'],d['ms_per_step']))"; done; done > $O/mb.txt 2>&1; cat $O/mb.txt
