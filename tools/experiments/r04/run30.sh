#!/bin/bash
O=gpurun_out/r04_run30; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nat.py -m gpu -q -x --timeout 600 > $O/pytest_nat.log 2>&1; echo "nat rc=$?"; grep -a "passed\|failed\|^E " $O/pytest_nat.log | cut -c1-300 | head
for rep in 1 2 3; do for e in 1 0; do
echo -n "no_early_encode=$e  "; if [ $e = 1 ]; then export VTTS_PIPE_NO_EARLY_ENCODE=1; else unset VTTS_PIPE_NO_EARLY_ENCODE; fi; timeout 300 python tools/pipeline_bench.py 256 1 3 x3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.endswith('_ms')})"; done; done
