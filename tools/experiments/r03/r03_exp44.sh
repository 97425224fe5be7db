#!/bin/bash
# the stage's ResBlocks side by side on parallel streams (chains = 2) on SOME stages only (env VTTS_CHAIN_STAGES = bit mask, experiment build)
O=gpurun_out/r03_exp44; mkdir -p $O
for r in 1 2; do for m in none 1 8 12 9 13 15; do
  echo -n "stages mask $m  "
  if [ $m = none ]; then CH=1; unset VTTS_CHAIN_STAGES; else CH=2; export VTTS_CHAIN_STAGES=$m; fi
  timeout 300 python bench.py --chains $CH --steps 6 --warmup 2 --no-cpu-baseline --no-f32 --no-rtf 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.4e ms %.2f roof %.3f'%(d['value'],d['ms_per_step'],d['roofline']['frac']))"
done; done 2>&1 | tee $O/ab.txt
