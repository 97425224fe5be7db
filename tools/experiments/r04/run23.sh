#!/bin/bash
# projection / prenet with several columns per thread (VTTS_NAT_PROJ_COLS=0: the generic kernel)
O=gpurun_out/r04_run23; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nat.py -m gpu -q -x --timeout 600 > $O/pytest_nat.log 2>&1; echo "nat rc=$?"; grep -a "acoustic model, bf16\|^\[text2mel\|pipeline (\|passed\|failed\|^E " $O/pytest_nat.log | cut -c1-300 | head
for cfg in "0 fp32" "1 fp32" "0 x3" "1 x3" "1 x3"; do set -- $cfg
echo -n "COLS=$1 $2  "; VTTS_NAT_PROJ_COLS=$1 timeout 300 python tools/pipeline_bench.py 256 1 3 $2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('acoustic_model_ms','total_ms')})"; done
cd /tmp && export TMPDIR=/tmp; R=/root/repo
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_pipe -o r -- python $R/tools/pipeline_bench.py 256 1 3 x3 > $R/$O/prof_pipe.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/$O/prof_pipe -name "*results.db" | head -1) $R/$O/prof_pipe_stats.md; grep "nat_dec" $R/$O/prof_pipe_stats.md | cut -c1-150 | head -5
find $R/$O -name "*.db" -delete
