#!/bin/bash
# split-state LSTM step: the cell state requested at kernel start + one-barrier reduction (each wave adds the K shares of its own block)
O=gpurun_out/r04_run26; mkdir -p $O
VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/libvtts_flat.so timeout 600 python -m pytest tests/test_gpu_nat.py -m gpu -q -x --timeout 300 -k "bf16x3 or reference_code_executed" 2>&1 | tail -2
for rep in 1 2; do for v in libvtts_hifigan.so libvtts_flat.so; do
echo -n "$v  "; VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/$v timeout 300 python tools/pipeline_bench.py 256 1 3 x3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('acoustic_model_ms','total_ms')})"; done; done
