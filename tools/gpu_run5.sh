mkdir -p gpurun_out; R=$PWD
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x --timeout 300 > gpurun_out/pytest_bf16.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_bf16.log
tail -4 gpurun_out/pytest_bf16.log
for mb in 0 2 8 16; do timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --microbatch $mb > gpurun_out/bench5_bf16_mb$mb.json 2> gpurun_out/bench5.err; python - <<PY
import json
d=json.load(open('gpurun_out/bench5_bf16_mb$mb.json'))
print('mb$mb', 'value %.3e'%d['value'], 'ms/step %.1f'%d['ms_per_step'], 'roof %.3f'%d['roofline']['frac'], 'rtf_ms %.3f'%d['rtf_b1']['latency_ms'])
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof5 -o r5 -- python $R/bench.py --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-rtf > $R/gpurun_out/prof5.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc5_sq -o sq -- python $R/bench.py --dtype bf16 --batch 4 --steps 1 --warmup 1 --no-cpu-baseline --no-rtf > $R/gpurun_out/pmc5_sq.log 2>&1
