#!/bin/bash
O=gpurun_out/r04_run4; mkdir -p $O
timeout 600 python tools/experiments/r04/diag_pipe3.py > $O/diag3.log 2>&1; echo "diag3 rc=$?"; cat $O/diag3.log | cut -c1-500
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "fp32_fused or golden or headline" > $O/pytest_f32.log 2>&1; echo "f32 rc=$?"; tail -2 $O/pytest_f32.log
for f in 0 2 3 0 2 3; do
  echo -n "fuse=$f " >> $O/f32_ab.log
  timeout 300 python bench.py --dtype f32 --fuse $f --steps 2 --warmup 1 --no-rtf --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.2f frac %.4f'%(d['ms_per_step'], d['frac_of_mfma_peak_whole_forward']))" >> $O/f32_ab.log
done
cat $O/f32_ab.log
