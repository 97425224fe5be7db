R=$PWD; mkdir -p gpurun_out
for cfg in "16 1" "32 1" "64 1" "32 2" "16 2"; do set -- $cfg
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline --no-rtf --steps 4 --microbatch $1 --streams $2 > gpurun_out/b12.json 2> gpurun_out/b12.err
python - <<PY
import json
d=json.load(open('gpurun_out/b12.json'))
print('mb/streams $cfg', 'value %.3e'%d['value'], 'ms/step %.1f'%d['ms_per_step'], 'roof %.3f'%d['roofline']['frac'], 'dominant avg_us %.1f'%(d['roofline']['avg_launch_ms']*1e3))
PY
done
