#!/bin/bash
O=gpurun_out/r04_run13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3.py -m gpu -q --timeout 600 > $O/pytest_x3.log 2>&1; echo "x3 rc=$?"; grep -a "ups_\|v1_scaled\|B=64\|passed\|failed\|^E " $O/pytest_x3.log | cut -c1-200 | head -24
for rep in 1 2; do
timeout 300 python bench.py --dtype bf16x3 --steps 3 --warmup 1 --no-rtf --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.2f (two streams)  calib one-stream %.2f  dominant %.3f ms'%(d['ms_per_step'], d['roofline']['calibration_ms_per_step'], d['roofline']['avg_launch_ms']))"
done
cd /tmp && export TMPDIR=/tmp; R=/root/repo
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_x3 -o r -- python $R/bench.py --dtype bf16x3 --steps 2 --warmup 1 --streams 1 --microbatch 64 --no-rtf --no-cpu-baseline > $R/$O/prof_x3.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/$O/prof_x3 -name "*results.db" | head -1) $R/$O/prof_x3_stats.md; head -26 $R/$O/prof_x3_stats.md | cut -c1-130
find $R/$O -name "*.db" -size +20M -delete
