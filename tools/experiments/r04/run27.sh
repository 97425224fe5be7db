#!/bin/bash
O=gpurun_out/r04_run27; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nat.py -m gpu -q -x --timeout 600 > $O/pytest_nat.log 2>&1; echo "nat rc=$?"; grep -a "acoustic model, bf16\|^\[text2mel\|with the bf16x3\|passed\|failed\|^E " $O/pytest_nat.log | cut -c1-200 | head -12
for cfg in fp32 x3 fp32 x3; do
echo -n "$cfg  "; timeout 300 python tools/pipeline_bench.py 256 1 3 $cfg 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('duration_model_ms','acoustic_model_ms','total_ms')})"; done
