R=$PWD; mkdir -p gpurun_out
for cfg in "bf16 0 1" "bf16 0 2" "bf16 0 3" "bf16 0 4" "bf16 2 2" "bf16 2 4" "bf16 8 2" "f32 0 1" "f32 0 2"; do set -- $cfg
timeout 300 python bench.py --dtype $1 --no-cpu-baseline --no-rtf --steps 4 --microbatch $2 --streams $3 > gpurun_out/b8.json 2> gpurun_out/b8.err
python - <<PY
import json
d=json.load(open('gpurun_out/b8.json'))
print('$cfg', 'value %.3e'%d['value'], 'ms/step %.1f'%d['ms_per_step'], 'dominant avg_us %.1f'%(d['roofline']['avg_launch_ms']*1e3))
PY
done
timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 2>&1 | tail -3
