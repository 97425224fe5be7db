#!/bin/bash
# stage-kernel development: per-variant duration of stage_bf16_k at 64 x 1024 frames (one-stream pass under rocprofv3), variants = libraries built by
#   python -m viettts_amd.csrc.build --define VTTS_ST_...=. --libname libvtts_<name>.so      usage: tools/r06_stage_variants.sh <tag> <libA.so> ...
T=$1; shift; R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 300 rocprofv3 --kernel-trace --stats -d $O/p_$v -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 --streams 1 --microbatch 64 > $O/p_$v.log 2>&1
  python $R/tools/rocprof_summary.py $(find $O/p_$v -name "*results.db" | head -1) $O/$v.md
  echo "$v: $(grep -E 'stage_bf16' $O/$v.md | cut -c1-120)  | $(grep 'all kernels' $O/$v.md)"
  rm -rf $O/p_$v
done | tee $O/variants.txt
