#!/bin/bash
O=gpurun_out/r03_exp27; mkdir -p $O; R=$PWD
timeout 1200 python -m pytest tests/test_gpu_nat.py -m gpu -q -x --timeout 900 2>&1 | tail -2
for r in 1 2 3; do for v in libvtts_prev.so libvtts_hifigan.so; do echo -n "$v "; VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/$v timeout 300 python tools/pipeline_bench.py 256 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dur %.2f ac %.2f gen %.2f total %.2f'%(d['duration_model_ms'],d['acoustic_model_ms'],d['generator_ms'],d['total_ms']))"; done; done
