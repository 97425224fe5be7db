#!/bin/bash
O=gpurun_out/r03_exp9; mkdir -p $O; R=$PWD
for sh in "32 11 3 16 262144" "32 7 5 16 262144" "32 3 1 16 262144" "32 11 5 2 1100" "32 3 1 2 700"; do for v in nowreg wreg; do echo "== $v $sh"; timeout 120 tools/kbench/bin/kbench_$v $sh 7 2>&1 | grep -a "time:\|check"; done; done > $O/kbench.txt 2>&1; grep -a "==\|time\|check" $O/kbench.txt | paste - - - | cut -c1-190
timeout 120 tools/kbench/bin/kbench_tlw 32 11 3 16 262144 3 20 2>&1 | grep -a "stage_x\|issue\|wait\|lrelu\|main loop\|epilogue\|add\|pack\|total"
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
bash tools/ab_bench.sh 3 libvtts_nowreg.so libvtts_hifigan.so libvtts_wreg96.so > $O/ab.txt 2>&1; cat $O/ab.txt
for v in libvtts_nowreg.so libvtts_hifigan.so; do echo -n "$v "; VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/$v timeout 300 python tools/pipeline_bench.py 256 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dur %.2f ac %.2f gen %.2f total %.2f'%(d['duration_model_ms'],d['acoustic_model_ms'],d['generator_ms'],d['total_ms']))"; done
