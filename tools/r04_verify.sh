#!/bin/bash
# full GPU verification of the current build: tests, smoke, counters of this very build (bf16, fp32, bf16x3 engines), bench line, and the kernel
# statistics of the 256-sentence text -> waveform pipeline with the acoustic model in fp32 and with its bf16x3 option
TAG=${1:-r04_i}
O=gpurun_out/r04_verify; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
bash tools/profile_final.sh $TAG > $O/profile_final.log 2>&1; tail -4 $O/profile_final.log
cp gpurun_out/$TAG/counters_bf16.json gpurun_out/$TAG/counters_f32.json gpurun_out/$TAG/counters_bf16x3.json profiles/ 2>/dev/null
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-700 $O/bench.json; tail -2 $O/bench.err
R=$PWD; cd /tmp && export TMPDIR=/tmp
for mode in fp32 x3; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/pipe_$mode -o r -- python $R/tools/pipeline_bench.py 256 3 $mode > $R/gpurun_out/$TAG/pipe_$mode.log 2>&1
  python $R/tools/rocprof_summary.py $(find $R/gpurun_out/$TAG/pipe_$mode -name "*results.db" | head -1) $R/gpurun_out/$TAG/${TAG}_pipeline_${mode}_kernel_stats.md
  tail -1 $R/gpurun_out/$TAG/pipe_$mode.log | cut -c1-300
done
find $R/gpurun_out/$TAG -name "*.db" -delete
