#!/bin/bash
O=gpurun_out/r04_run9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3.py -m gpu -q --timeout 600 > $O/pytest_x3.log 2>&1; echo "x3 rc=$?"; grep -a "^\[bf16x3\|passed\|failed\|Error\|^E " $O/pytest_x3.log | cut -c1-250 | head -70
timeout 600 python bench.py --dtype bf16x3 --steps 3 --warmup 1 --no-rtf --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.err; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r04_run9/bench_x3.json').read().strip().splitlines()[-1])
    print('x3 ms/step %.2f value %.4e'%(d['ms_per_step'], d['value']), 'roof', d['roofline'] and {k:d['roofline'][k] for k in ('achieved','avg_launch_ms','kernel')})
except Exception as e: print('bench parse failed', e)
PY
tail -3 $O/bench_x3.err
cd /tmp && export TMPDIR=/tmp; R=/root/repo
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_x3 -o r -- python $R/bench.py --dtype bf16x3 --steps 1 --warmup 1 --streams 1 --microbatch 64 --no-rtf --no-cpu-baseline > $R/$O/prof_x3.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/$O/prof_x3 -name "*results.db" | head -1) $R/$O/prof_x3_stats.md; head -24 $R/$O/prof_x3_stats.md | cut -c1-160
find $R/$O -name "*.db" -size +20M -delete
