#!/bin/bash
# whole-ResBlock split kernel: start-up skew of the second batch of workgroups (kernels_x3_rb.hip), per-kernel durations for a few settings
TAG=${1:-r05_skew}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for sk in ${SKEWS:-"0,0,0" "2,4,6" "3,5,7" "5,8,11"}; do
  n=$(echo $sk | tr ',' '_')
  VTTS_RX_SKEW=$sk timeout 600 rocprofv3 --kernel-trace --stats -d $O/s$n -o r -- python $R/bench.py --dtype bf16x3 --fuse 3 --streams 1 --microbatch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 > $O/s$n.log 2>&1
  python $R/tools/rocprof_summary.py $(find $O/s$n -name "*results.db" | head -1) $O/skew${n}_kernel_stats.md
  echo "== skew $sk"; grep -E "resblock_x3_k|all kernels" $O/skew${n}_kernel_stats.md | cut -c1-140
done
find $O -name "*.db" -delete
