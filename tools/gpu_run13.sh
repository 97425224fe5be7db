R=$PWD; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/bench13.json 2> gpurun_out/bench13.err; cat gpurun_out/bench13.json; tail -2 gpurun_out/bench13.err
timeout 600 python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/bench13_f32.json 2> gpurun_out/bench13_f32.err; cat gpurun_out/bench13_f32.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof13 -o r13 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 > $R/gpurun_out/prof13.log 2>&1
