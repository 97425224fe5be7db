#!/bin/bash
# round 4, GPU call 1: the new oracle-level tests on the bench's own legs, the NAT changes (hoisted gates, grouped hand-over), the
# overlapped pipeline (A/B over the number of groups), the two-stream bf16 default + calibration pass
O=gpurun_out/r04_run1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nat.py -m gpu -q -x --timeout 600 > $O/pytest_nat.log 2>&1; echo "nat rc=$?"; tail -4 $O/pytest_nat.log
timeout 900 python -m pytest tests/test_gpu_longform.py tests/test_gpu_dist.py "tests/test_gpu_parity.py::test_fp32_headline_shape_B64_T1024_default_schedule" -m gpu -q -x --timeout 600 > $O/pytest_new.log 2>&1; echo "new rc=$?"; grep -a "^\[" $O/pytest_new.log; tail -3 $O/pytest_new.log
for og in 1 2 3 4 6 8 1 4; do
  echo -n "overlap_groups=$og " >> $O/pipe_ab.log
  timeout 300 python tools/pipeline_bench.py 256 $og 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.endswith('_ms') or k=='overlap_groups'})" >> $O/pipe_ab.log
done
cat $O/pipe_ab.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_run1/bench.json'))
print('value %.4e ms/step %.2f'%(d['value'],d['ms_per_step']), 'roof', d['roofline']['frac'], 'calib ms', d['roofline'].get('calibration_ms_per_step'))
print('fp32', {k:d['fp32_path'][k] for k in ('samples_per_s','frac_of_f32_mfma_peak','parity','b1_T512_latency_ms')})
print('pipe', d['pipeline_256']); print('long', d['longform_10min']); print('rtf', d['rtf_b1'])
PY
tail -3 $O/bench.err
