#!/bin/bash
# fp32 MFMA convolutions on wider time tiles (less halo per output, two workgroups per CU instead of three)
O=gpurun_out/r03_exp45; mkdir -p $O; R=$PWD
for v in "$@"; do
VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 2>&1 | tail -1
done
for r in 1 2; do for v in "$@"; do
  echo -n "$v  "
  VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 300 python bench.py --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-rtf 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.4e ms %.2f'%(d['value'],d['ms_per_step']))"
done; done 2>&1 | tee $O/ab.txt
