R=$PWD; mkdir -p gpurun_out
for v in hifigan abl_NOB abl_NOA abl_NODMA abl_NOBAR abl_NOMFMA; do
VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/libvtts_$v.so timeout 120 python bench.py --dtype bf16 --no-cpu-baseline --no-rtf --steps 3 > gpurun_out/b10_$v.json 2> gpurun_out/b10.err
python - <<PY
import json
try:
    d=json.load(open('gpurun_out/b10_$v.json'))
    print('%-12s'%'$v', 'ms/step %.1f'%d['ms_per_step'], 'C128k11 pair avg_us %.1f'%(d['roofline']['avg_launch_ms']*1e3))
except Exception as e: print('$v failed', e)
PY
done
