#!/bin/bash
# split-operand engine, 64 x 1024 frames: per-kernel durations (rocprofv3 --kernel-trace --stats) under fuse = 1 (pairs only), 2 (default policy),
# 3 (the whole-ResBlock kernel wherever it exists), one stream, one pass of 64 — the A/B behind engine.hip: resblock_x3_preferred
TAG=${1:-r05_x3ab}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for f in ${FUSES:-1 2 3}; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/f$f -o r -- python $R/bench.py --dtype bf16x3 --fuse $f --streams 1 --microbatch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 > $O/f$f.log 2>&1
  python $R/tools/rocprof_summary.py $(find $O/f$f -name "*results.db" | head -1) $O/fuse${f}_kernel_stats.md
  tail -1 $O/f$f.log | cut -c1-200
done
find $O -name "*.db" -delete
