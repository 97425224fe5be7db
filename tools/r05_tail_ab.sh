#!/bin/bash
# bf16 engine, 64 x 1024 frames: option "tail" on / off, interleaved three times on one box
for rep in 1 2 3; do for t in 1 0; do
  python bench.py --tail $t --no-cpu-baseline --no-f32 --no-rtf --steps 20 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail', $t, 'ms_per_step', round(d['ms_per_step'],3), 'calib', round(d['roofline']['calibration_ms_per_step'],3))"
done; done
