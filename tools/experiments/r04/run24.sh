#!/bin/bash
# whole-ResBlock kernel at C = 128, k = 3: a 256-step window on eight waves (one workgroup per CU) against the 128-step one and against three pair launches
O=gpurun_out/r04_run24; mkdir -p $O
R=$PWD
VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/libvtts_rb128w.so timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x --timeout 300 -k "golden and fused-resblocks-all" > $O/pytest.log 2>&1; echo "golden (fuse=3, wide window) rc=$?"; tail -2 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for v in libvtts_hifigan.so libvtts_rb128w.so; do
VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$v -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 --streams 1 --microbatch 64 --fuse 3 > $R/$O/prof_$v.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/$O/prof_$v -name "*results.db" | head -1) $R/$O/prof_$v.md; echo "== $v (fuse 3)"; grep "RBTile<128\|GTile<128, 3" $R/$O/prof_$v.md | cut -c1-150
done
find $R/$O -name "*.db" -delete
