#!/bin/bash
O=gpurun_out/r03_exp10; mkdir -p $O; R=$PWD
for v in libvtts_nowreg.so libvtts_hifigan.so; do
  (cd /tmp && export TMPDIR=/tmp && VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_$v -o r -- python $R/bench.py --fuse 3 --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 > $R/$O/trace_$v.log 2>&1)
  python tools/rocprof_summary.py $(find $O/trace_$v -name "*results.db" | head -1) $O/stats_$v.md; echo "== $v"; grep "resblock_bf16_k\|GTile<32" $O/stats_$v.md | cut -c1-130
done
find $O -name "*.db" -size +20M -delete
