mkdir -p gpurun_out
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -q -v --timeout 300 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; echo "bench rc=$?" >> gpurun_out/bench1.err
R=$PWD; cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof1 -o r1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rtf > $R/gpurun_out/prof1.log 2>&1; echo "prof rc=$?" >> $R/gpurun_out/prof1.log
cd $R; tail -5 gpurun_out/smoke.log; tail -40 gpurun_out/pytest_gpu.log; cat gpurun_out/bench1.json; tail -5 gpurun_out/bench1.err; ls -R gpurun_out/prof1 | head
