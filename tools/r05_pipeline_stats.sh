#!/bin/bash
# kernel statistics of the 256-sentence pipeline in its throughput (x3) and its parity-grade configuration
TAG=${1:-r05_g}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in x3 parity; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/pipe_$mode -o r -- python $R/tools/pipeline_bench.py 256 3 $mode > $O/pipe_$mode.log 2>&1
  python $R/tools/rocprof_summary.py $(find $O/pipe_$mode -name "*results.db" | head -1) $O/${TAG}_pipeline_${mode}_kernel_stats.md --note "rocprofv3 --kernel-trace --stats -- python tools/pipeline_bench.py 256 3 $mode (three passes of the 256-sentence pipeline; mode: tools/pipeline_bench.py)"
  tail -1 $O/pipe_$mode.log | cut -c1-400
done
find $O -name "*.db" -delete
