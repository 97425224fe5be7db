#!/bin/bash
O=gpurun_out/r04_run10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3.py -m gpu -q --timeout 600 > $O/pytest_x3.log 2>&1; echo "x3 rc=$?"; grep -a "v1_scaled\|B=64\|passed\|failed\|^E " $O/pytest_x3.log | cut -c1-200 | head
cd /tmp && export TMPDIR=/tmp; R=/root/repo
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_x3 -o r -- python $R/bench.py --dtype bf16x3 --steps 2 --warmup 1 --streams 1 --microbatch 64 --no-rtf --no-cpu-baseline > $R/$O/prof_x3.log 2>&1
grep -a "ms_per_step" $R/$O/prof_x3.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x3 one-stream ms/step %.2f'%d['ms_per_step'])"
python $R/tools/rocprof_summary.py $(find $R/$O/prof_x3 -name "*results.db" | head -1) $R/$O/prof_x3_stats.md; grep "x3_k" $R/$O/prof_x3_stats.md | cut -c1-140
find $R/$O -name "*.db" -size +20M -delete
