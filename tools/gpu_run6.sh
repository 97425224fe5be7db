mkdir -p gpurun_out; R=$PWD
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x --timeout 300 > gpurun_out/pytest_bf16.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_bf16.log
tail -4 gpurun_out/pytest_bf16.log
for mb in 0 8; do timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --microbatch $mb > gpurun_out/bench6_bf16_mb$mb.json 2> gpurun_out/bench6.err; python - <<PY
import json
d=json.load(open('gpurun_out/bench6_bf16_mb$mb.json'))
print('mb$mb', 'value %.3e'%d['value'], 'ms/step %.1f'%d['ms_per_step'], 'roof %.3f'%d['roofline']['frac'], 'rtf_ms %.3f'%d['rtf_b1']['latency_ms'], d['roofline']['kernel'])
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof6 -o r6 -- python $R/bench.py --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-rtf > $R/gpurun_out/prof6.log 2>&1
