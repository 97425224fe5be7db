#!/bin/bash
# fp32 engine at the headline shape: the stage's ResBlocks on parallel streams (chains = 2) and micro-batch / stream splits
O=gpurun_out/r03_exp35; mkdir -p $O
for r in 1 2; do for cfg in "1 0 1" "2 0 1" "1 32 2" "1 16 2" "1 16 4"; do set -- $cfg; echo -n "chains $1 microbatch $2 streams $3  "; timeout 300 python bench.py --dtype f32 --chains $1 --microbatch $2 --streams $3 --steps 4 --warmup 1 --no-cpu-baseline --no-rtf 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.4e ms %.2f'%(d['value'],d['ms_per_step']))"; done; done > $O/f32.txt 2>&1; cat $O/f32.txt
