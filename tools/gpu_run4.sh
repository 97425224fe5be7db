mkdir -p gpurun_out; R=$PWD
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x --timeout 300 > gpurun_out/pytest_bf16.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_bf16.log
tail -40 gpurun_out/pytest_bf16.log
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/bench4_bf16.json 2> gpurun_out/bench4_bf16.err; echo "rc=$?"; tail -3 gpurun_out/bench4_bf16.err; cat gpurun_out/bench4_bf16.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof4 -o r4 -- python $R/bench.py --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-rtf > $R/gpurun_out/prof4.log 2>&1
