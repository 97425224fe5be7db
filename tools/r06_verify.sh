#!/bin/bash
# full GPU verification of the current build (round 6): tests, smoke, counters of this very build (bf16, fp32, bf16x3 engines), the bench line,
# and the kernel statistics of the 256-sentence pipeline in its throughput and its parity-grade configuration.  gpurun --timeout 3000 -- 'bash tools/r06_verify.sh r06_g'
TAG=${1:-r06_g}
O=gpurun_out/r06_verify; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
bash tools/profile_final.sh $TAG > $O/profile_final.log 2>&1; tail -4 $O/profile_final.log
cp gpurun_out/$TAG/counters_bf16.json gpurun_out/$TAG/counters_f32.json gpurun_out/$TAG/counters_bf16x3.json profiles/ 2>/dev/null
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-700 $O/bench.json; tail -2 $O/bench.err
bash tools/r05_pipeline_stats.sh $TAG 2>&1 | tail -4
