# Quick GPU check: bf16 + long-form + NAT tests, a bench line, rocprofv3 kernel stats.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_longform.py tests/test_gpu_nat.py -m gpu -q -x --timeout 300 2>&1 | tail -5
timeout 600 python bench.py --no-cpu-baseline --no-f32 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_quick.json'))
print('value %.4e'%d['value'], 'ms/step %.2f'%d['ms_per_step'], 'roof %.3f'%d['roofline']['frac'], 'rtf_ms %.3f'%d['rtf_b1']['latency_ms'])
PY
cd /tmp && export TMPDIR=/tmp; R=/root/repo
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_quick -o rq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 > $R/gpurun_out/prof_quick.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_quick -name "*results.db" | head -1) $R/gpurun_out/prof_quick_stats.md; head -24 $R/gpurun_out/prof_quick_stats.md
