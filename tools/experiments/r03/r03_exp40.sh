#!/bin/bash
# consecutive launches walk the batch in alternating directions (env VTTS_ZREV=1): the consumer starts with the utterances the producer wrote last (Infinity Cache)
O=gpurun_out/r03_exp40; mkdir -p $O
# (the env switch of this experiment became the engine option "zigzag", default 1)
VTTS_ZREV=1 timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x --timeout 600 -k "golden or ragged or invariance or edge_lengths" 2>&1 | tail -1
for r in 1 2 3; do for z in 0 1; do
  echo -n "zrev $z  "
  if [ $z = 1 ]; then export VTTS_ZREV=1; else unset VTTS_ZREV; fi
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-f32 --no-rtf 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.4e ms %.2f roof %.3f'%(d['value'],d['ms_per_step'],d['roofline']['frac']))"
done; done 2>&1 | tee $O/ab.txt
