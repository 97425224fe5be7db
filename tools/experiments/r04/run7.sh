#!/bin/bash
O=gpurun_out/r04_run7; mkdir -p $O
tools/kbench/bin/pkfma_hazard 20000 2>&1 | tee $O/pkfma_hazard.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 > $O/pytest_f32.log 2>&1; echo "f32 rc=$?"; tail -2 $O/pytest_f32.log
run() { echo -n "$1 fuse=$2 " >> $O/f32_ab.log
  VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/$1 timeout 300 python bench.py --dtype f32 --fuse $2 --steps 2 --warmup 1 --no-rtf --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.2f frac %.4f roof %.4f'%(d['ms_per_step'], d['frac_of_mfma_peak_whole_forward'], d['roofline']['frac']))" >> $O/f32_ab.log; }
for rep in 1 2; do run libvtts_pf1.so 2; run libvtts_hifigan.so 2; run libvtts_pf1.so 0; run libvtts_hifigan.so 0; done
cat $O/f32_ab.log
