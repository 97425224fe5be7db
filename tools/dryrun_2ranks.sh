#!/bin/bash
# N > 1 code path of bench.py on a one-GPU box: 2 ranks share cuda:0, gloo collectives (development check, not a measurement)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
VTTS_DIST_BACKEND=gloo VTTS_SHARE_GPU=1 timeout 600 python \
  bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-f32 > gpurun_out/dryrun2.json 2> gpurun_out/dryrun2.err
echo "rc=$?"; tail -3 gpurun_out/dryrun2.err | cut -c1-300; python - <<'PY'
import json
d=json.loads(open('gpurun_out/dryrun2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('n_gpus','value','ms_per_step')}); print(d.get('pipeline_256'))
PY
