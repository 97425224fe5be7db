#!/bin/bash
# three workgroups per CU where the LDS tile allows it: C = 64, k = 7 pairs on 384-step tiles; C = 128, k = 3 pairs on 192-step tiles; the C = 32, k = 7 whole-ResBlock kernel on 384-step windows
O=gpurun_out/r03_exp33; mkdir -p $O; R=$PWD
for v in libvtts_hifigan.so libvtts_g64k7.so libvtts_g128k3.so libvtts_rb32k7.so; do
VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x --timeout 600 -k "fused_pair_kat or golden or fused_resblock_equals or edge_lengths" 2>&1 | tail -1
(cd /tmp && export TMPDIR=/tmp && VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_$v -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 > $R/$O/trace_$v.log 2>&1)
python tools/rocprof_summary.py $(find $O/trace_$v -name "*results.db" | head -1) $O/stats_$v.md; echo "== $v"; grep -E "GTile<64, 7|GTile<128, 3|RBTile<32, 7" $O/stats_$v.md | cut -c1-150
done
find $O -name "*.db" -size +20M -delete
bash tools/ab_bench.sh 2 libvtts_hifigan.so libvtts_g64k7.so libvtts_g128k3.so libvtts_rb32k7.so
