#!/bin/bash
# activation rows leave the bf16 kernels as non-temporal stores (-DVTTS_NT_STORE=1): L2 kept for the rows that are re-read (halo, residual)
O=gpurun_out/r03_exp37; mkdir -p $O; R=$PWD
V=${1:-libvtts_nt.so}
for v in libvtts_hifigan.so $V; do
VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x --timeout 600 -k "kat or golden or fused_resblock_equals or edge_lengths or ragged" 2>&1 | tail -1
(cd /tmp && export TMPDIR=/tmp && VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_$v -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 > $R/$O/trace_$v.log 2>&1)
python tools/rocprof_summary.py $(find $O/trace_$v -name "*results.db" | head -1) $O/stats_$v.md; echo "== $v"; grep -E "_k<" $O/stats_$v.md | cut -c1-150 | head -22
done
find $O -name "*.db" -size +20M -delete
bash tools/ab_bench.sh 3 libvtts_hifigan.so $V
