#!/bin/bash
O=gpurun_out/r04_run8; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_run8/bench.json'))
print('value %.4e ms/step %.2f'%(d['value'],d['ms_per_step']), 'roof', d['roofline']['frac'], 'calib ms', d['roofline'].get('calibration_ms_per_step'), d['roofline'].get('counters_unavailable'))
print('fp32', {k:d['fp32_path'][k] for k in ('samples_per_s','frac_of_f32_mfma_peak','b1_T512_latency_ms')}, d['fp32_path']['parity']['max_abs_wav_vs_fp64_reference'])
print('pipe', {k:v for k,v in d['pipeline_256'].items() if k!='workload'}); print('long', {k:v for k,v in d['longform_10min'].items() if k!='workload'}); print('rtf', d['rtf_b1']['latency_ms'], d['rtf_b1']['latency_ms_eager_launches'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
tail -2 $O/bench.err
