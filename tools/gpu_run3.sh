mkdir -p gpurun_out; R=$PWD
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench3.json 2> gpurun_out/bench3.err
VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/libvtts_exp_pin.so timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench3_pin.json 2> gpurun_out/bench3_pin.err
timeout 600 python bench.py --no-cpu-baseline --microbatch 8 > gpurun_out/bench3_mb8.json 2> gpurun_out/bench3_mb8.err
timeout 600 python bench.py --no-cpu-baseline --microbatch 16 > gpurun_out/bench3_mb16.json 2> gpurun_out/bench3_mb16.err
tail -5 gpurun_out/pytest_gpu.log; for f in bench3 bench3_pin bench3_mb8 bench3_mb16; do python - <<PY
import json
d=json.load(open('gpurun_out/$f.json'))
print('$f', 'value %.3e'%d['value'], 'ms/step %.1f'%d['ms_per_step'], 'roof %.3f'%d['roofline']['frac'], 'rtf_ms %.3f'%d['rtf_b1']['latency_ms'])
PY
done
