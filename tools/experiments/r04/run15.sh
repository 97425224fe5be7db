#!/bin/bash
# NAT decoder step: two slices per workgroup in the LSTM step (VTTS_NAT_SL) and the loads-ahead projection / prenet kernel (VTTS_NAT_AHEAD), A/B
O=gpurun_out/r04_run15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nat.py -m gpu -q -x --timeout 600 > $O/pytest_nat.log 2>&1; echo "nat rc=$?"; grep -a "^\[text2mel\|passed\|failed\|^E " $O/pytest_nat.log | cut -c1-200 | head
for cfg in "1 0" "2 0" "1 1" "2 1" "2 1"; do set -- $cfg
echo "SL=$1 AHEAD=$2"; VTTS_NAT_SL=$1 VTTS_NAT_AHEAD=$2 timeout 300 python tools/pipeline_bench.py 256 1 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.endswith('_ms')})"; done
cd /tmp && export TMPDIR=/tmp; R=/root/repo
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_pipe -o r -- python $R/tools/pipeline_bench.py 256 1 3 > $R/$O/prof_pipe.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/$O/prof_pipe -name "*results.db" | head -1) $R/$O/prof_pipe_stats.md; grep "nat_" $R/$O/prof_pipe_stats.md | cut -c1-150
find $R/$O -name "*.db" -size +20M -delete
