#!/bin/bash
# interleaved A/B/... of several builds of the library on the same box: tools/ab_bench.sh <rounds> <libA.so> <libB.so> [...]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/ab.log
: > $O
R=$1; shift
for r in $(seq 1 $R); do for v in "$@"; do
  echo -n "$v  " >> $O
  VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-f32 --no-rtf 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.4e ms %.2f roof %.3f'%(d['value'],d['ms_per_step'],d['roofline']['frac']))" >> $O
done; done
cat $O
