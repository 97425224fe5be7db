#!/bin/bash
O=gpurun_out/r04_run11; mkdir -p $O
for lib in libvtts_hifigan.so libvtts_x3b.so; do
echo "== $lib"
VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/$lib timeout 900 python -m pytest tests/test_gpu_x3.py -m gpu -q --timeout 600 > $O/pytest_$lib.log 2>&1; echo "x3 rc=$?"; grep -a "B=64\|passed\|failed\|^E " $O/pytest_$lib.log | cut -c1-200 | head -5
for rep in 1 2; do
VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/$lib timeout 300 python bench.py --dtype bf16x3 --steps 3 --warmup 1 --no-rtf --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.2f (two streams)  calib one-stream %.2f  dominant %.3f ms'%(d['ms_per_step'], d['roofline']['calibration_ms_per_step'], d['roofline']['avg_launch_ms']))"
done
done
cd /tmp && export TMPDIR=/tmp; R=/root/repo
for lib in libvtts_hifigan.so libvtts_x3b.so; do
VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$lib timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$lib -o r -- python $R/bench.py --dtype bf16x3 --steps 2 --warmup 1 --streams 1 --microbatch 64 --no-rtf --no-cpu-baseline > $R/$O/prof_$lib.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/$O/prof_$lib -name "*results.db" | head -1) $R/$O/prof_${lib}_stats.md; echo "== $lib"; grep "x3_k" $R/$O/prof_${lib}_stats.md | cut -c1-120
done
find $R/$O -name "*.db" -size +20M -delete
