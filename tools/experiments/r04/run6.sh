#!/bin/bash
O=gpurun_out/r04_run6; mkdir -p $O
tools/kbench/bin/pkfma_hazard 20000 2>&1 | tee $O/pkfma_hazard.txt
timeout 900 python -m pytest tests/test_gpu_nat.py -m gpu -q -x --timeout 600 > $O/pytest_nat.log 2>&1; echo "nat rc=$?"; grep -a "^\[" $O/pytest_nat.log; tail -3 $O/pytest_nat.log
timeout 300 python tools/experiments/r04/diag_pipe3.py short 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tee $O/diag_fixed.log
for og in 1 4 6 1 4 6; do
  echo -n "overlap_groups=$og " >> $O/pipe_ab.log
  timeout 300 python tools/pipeline_bench.py 256 $og 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.endswith('_ms') or k=='overlap_groups'})" >> $O/pipe_ab.log
done
cat $O/pipe_ab.log
for f in 0 2; do
  echo -n "fuse=$f " >> $O/f32_ab.log
  timeout 300 python bench.py --dtype f32 --fuse $f --steps 2 --warmup 1 --no-rtf --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.2f frac %.4f'%(d['ms_per_step'], d['frac_of_mfma_peak_whole_forward']))" >> $O/f32_ab.log
done
cat $O/f32_ab.log
