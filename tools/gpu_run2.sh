# GPU run 2: parity of the new MFMA convT / conv_pre / conv_post kernels, bench, rocprof stats + PMC passes
mkdir -p gpurun_out; R=$PWD
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench2.json 2> gpurun_out/bench2.err; echo "bench rc=$?" >> gpurun_out/bench2.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof2 -o r2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rtf > $R/gpurun_out/prof2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof2_b1 -o b1 -- python $R/bench.py --batch 1 --frames 512 --steps 5 --warmup 2 --no-cpu-baseline --no-rtf > $R/gpurun_out/prof2_b1.log 2>&1
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc2_sq -o sq -- python $R/bench.py --batch 4 --steps 1 --warmup 1 --no-cpu-baseline --no-rtf > $R/gpurun_out/pmc2_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc2_fetch -o f -- python $R/bench.py --batch 4 --steps 1 --warmup 1 --no-cpu-baseline --no-rtf > $R/gpurun_out/pmc2_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc2_write -o w -- python $R/bench.py --batch 4 --steps 1 --warmup 1 --no-cpu-baseline --no-rtf > $R/gpurun_out/pmc2_write.log 2>&1
cd $R; tail -3 gpurun_out/smoke.log; tail -15 gpurun_out/pytest_gpu.log; cat gpurun_out/bench2.json; tail -3 gpurun_out/bench2.err; ls gpurun_out/*
