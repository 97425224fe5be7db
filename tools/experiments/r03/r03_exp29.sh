#!/bin/bash
O=gpurun_out/r03_exp29; mkdir -p $O; R=$PWD
for v in libvtts_hifigan.so libvtts_g64wm2.so; do
VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x --timeout 600 -k "fused_pair_kat or golden" 2>&1 | tail -1
(cd /tmp && export TMPDIR=/tmp && VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_$v -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 > $R/$O/trace_$v.log 2>&1)
python tools/rocprof_summary.py $(find $O/trace_$v -name "*results.db" | head -1) $O/stats_$v.md; echo "== $v"; grep "GTile<64, " $O/stats_$v.md | cut -c1-130
done
find $O -name "*.db" -size +20M -delete
