cd tools/kbench
for cfg in "256 3 1 16 8192" "256 7 3 16 8192" "256 11 5 16 8192" "256 11 1 64 8192"; do
    echo "== cfg $cfg impl 30: $(timeout 60 ./kbench $cfg 7 30 | grep -E '^time|check' | tr '\n' ' ')"
done
timeout 60 ./kbench_tl 256 11 5 64 8192 3 30 | grep -A7 "^timeline"
cd ../..
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x --timeout 300 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-f32 > gpurun_out/bench15.json 2> gpurun_out/bench15.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench15.json'))
print('value %.4e'%d['value'], 'ms/step %.2f'%d['ms_per_step'], 'roof %.3f'%d['roofline']['frac'], 'rtf_ms %.3f'%d['rtf_b1']['latency_ms'])
PY
