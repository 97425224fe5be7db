R=$PWD; mkdir -p gpurun_out; export VTTS_BENCH_ALLOW_GARBAGE=1
for v in hifigan exp_regs abl_NODMA abl_NOBAR; do
VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/libvtts_$v.so timeout 120 python bench.py --dtype bf16 --no-cpu-baseline --no-rtf --steps 3 > gpurun_out/b11_$v.json 2> gpurun_out/b11_$v.err
python - <<PY
import json
try:
    d=json.load(open('gpurun_out/b11_$v.json'))
    print('%-12s'%'$v', 'ms/step %.1f'%d['ms_per_step'], 'C128k11 pair avg_us %.1f'%(d['roofline']['avg_launch_ms']*1e3))
except Exception as e: print('$v failed', e); print(open('gpurun_out/b11_$v.err').read()[-400:])
PY
done
VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/libvtts_exp_regs.so timeout 300 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x --timeout 120 2>&1 | tail -3
