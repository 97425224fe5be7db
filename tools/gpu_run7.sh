R=$PWD; mkdir -p gpurun_out
for v in hifigan exp_NOSTORE exp_NOXLOAD exp_NOMFMA exp_NOSLAB; do
VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/libvtts_$v.so timeout 300 python bench.py --dtype bf16 --no-cpu-baseline --no-rtf --steps 3 > gpurun_out/b7_$v.json 2> gpurun_out/b7.err
python - <<PY
import json
d=json.load(open('gpurun_out/b7_$v.json'))
print('%-14s'%'$v', 'ms/step %.1f'%d['ms_per_step'], 'dominant avg_us %.1f'%(d['roofline']['avg_launch_ms']*1e3))
PY
done
