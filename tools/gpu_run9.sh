R=$PWD; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 300 > gpurun_out/pytest_bf16.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_bf16.log
tail -25 gpurun_out/pytest_bf16.log
for cfg in "0 1" "0 2" "2 2" "8 1"; do set -- $cfg
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline --steps 4 --microbatch $1 --streams $2 > gpurun_out/b9.json 2> gpurun_out/b9.err
python - <<PY
import json
d=json.load(open('gpurun_out/b9.json'))
print('mb/streams $cfg', 'value %.3e'%d['value'], 'ms/step %.1f'%d['ms_per_step'], 'roof %.3f'%d['roofline']['frac'], 'dominant avg_us %.1f'%(d['roofline']['avg_launch_ms']*1e3), 'rtf_ms %.3f'%d['rtf_b1']['latency_ms'])
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof9 -o r9 -- python $R/bench.py --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-rtf > $R/gpurun_out/prof9.log 2>&1
