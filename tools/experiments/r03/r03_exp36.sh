#!/bin/bash
# epilogue 2 of the C <= 64 pairs through LDS, row-contiguous stores (-DVTTS_STAGE_Y=1) vs 16-byte stores from the accumulator layout
O=gpurun_out/r03_exp36; mkdir -p $O; R=$PWD
V=${1:-libvtts_stagey.so}
for v in libvtts_hifigan.so $V; do
VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x --timeout 600 -k "fused_pair_kat or golden or fused_resblock_equals or edge_lengths or ragged" 2>&1 | tail -1
(cd /tmp && export TMPDIR=/tmp && VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_$v -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 > $R/$O/trace_$v.log 2>&1)
python tools/rocprof_summary.py $(find $O/trace_$v -name "*results.db" | head -1) $O/stats_$v.md; echo "== $v"; grep -E "GTile<64, |GTile<32, " $O/stats_$v.md | cut -c1-150
done
find $O -name "*.db" -size +20M -delete
bash tools/ab_bench.sh 3 libvtts_hifigan.so $V
