#!/bin/bash
O=gpurun_out/r04_run22; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nat.py -m gpu -q -x --timeout 600 > $O/pytest_nat.log 2>&1; echo "nat rc=$?"; grep -a "acoustic model, bf16\|^\[text2mel\|passed\|failed\|^E " $O/pytest_nat.log | cut -c1-300 | head
for cfg in fp32 x3 x3; do
echo -n "$cfg  "; timeout 300 python tools/pipeline_bench.py 256 1 3 $cfg 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.endswith('_ms') or k == 'acoustic_precision'})"; done
cd /tmp && export TMPDIR=/tmp; R=/root/repo
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_pipe -o r -- python $R/tools/pipeline_bench.py 256 1 3 x3 > $R/$O/prof_pipe.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/$O/prof_pipe -name "*results.db" | head -1) $R/$O/prof_pipe_stats.md; grep "nat_" $R/$O/prof_pipe_stats.md | cut -c1-150 | head -12
find $R/$O -name "*.db" -delete
