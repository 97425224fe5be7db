#!/bin/bash
# bf16 engine, 64 x 1024 frames: option "stage" (the whole last stage in one launch) on / off, interleaved three times on one box, then the
# per-kernel durations of a one-stream pass with the option on (rocprofv3 --kernel-trace --stats).   gpurun -- 'bash tools/r06_stage_ab.sh [tag]'
T=${1:-r06_stage}; R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O
for rep in 1 2 3; do for t in 1 0; do
  python bench.py --stage $t --no-cpu-baseline --no-f32 --no-rtf --steps 20 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stage', $t, 'ms_per_step', round(d['ms_per_step'],3), 'calib', round(d['roofline']['calibration_ms_per_step'],3))"
done; done | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for t in 1 0; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof$t -o r -- python $R/bench.py --stage $t --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 --streams 1 --microbatch 64 > $O/prof$t.log 2>&1
  python $R/tools/rocprof_summary.py $(find $O/prof$t -name "*results.db" | head -1) $O/stage${t}_kernel_stats.md
  grep -E "stage_bf16|RBTile<32|GTile<32|GTail|UTile<64, 64|conv_post|all kernels" $O/stage${t}_kernel_stats.md | cut -c1-150
done
find $O -name "*.db" -size +20M -delete
