#!/bin/bash
O=gpurun_out/r03_exp20; mkdir -p $O; R=$PWD
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log | cut -c1-300
bash tools/ab_bench.sh 3 libvtts_prev.so libvtts_hifigan.so > $O/ab.txt 2>&1; cat $O/ab.txt
for v in libvtts_prev.so libvtts_hifigan.so; do echo -n "$v "; VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/$v timeout 300 python tools/pipeline_bench.py 256 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gen %.2f total %.2f'%(d['generator_ms'],d['total_ms']))"; done
