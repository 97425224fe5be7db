#!/bin/bash
# round 4, GPU call 3: pipeline race diagnostic 2; fp32 fused pairs after the latency work (ring of 4 A iterations, B fragments a k-step ahead,
# batched residual loads in both fp32 kernels' epilogues): tests, fuse levels, tile-geometry variants
O=gpurun_out/r04_run3; mkdir -p $O
timeout 600 python tools/experiments/r04/diag_pipe2.py 12 > $O/diag2.log 2>&1; echo "diag2 rc=$?"; grep -v "^[0-9]* nfr" $O/diag2.log | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 > $O/pytest_f32.log 2>&1; echo "f32 rc=$?"; tail -3 $O/pytest_f32.log
run() { # lib fuse
  echo -n "$1 fuse=$2 " >> $O/f32_ab.log
  VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/$1 timeout 300 python bench.py --dtype f32 --fuse $2 --steps 2 --warmup 1 --no-rtf --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.2f frac %.4f'%(d['ms_per_step'], d['frac_of_mfma_peak_whole_forward']))" >> $O/f32_ab.log
}
for rep in 1 2; do
run libvtts_hifigan.so 0; run libvtts_hifigan.so 2; run libvtts_hifigan.so 3
run libvtts_f64n64.so 2; run libvtts_f64n256.so 2; run libvtts_f32n512.so 2; run libvtts_f32n128.so 2
done
cat $O/f32_ab.log
cd /tmp && export TMPDIR=/tmp; R=/root/repo
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_f32 -o r -- python $R/bench.py --dtype f32 --fuse 3 --steps 1 --warmup 1 --streams 1 --no-rtf --no-cpu-baseline > $R/$O/prof_f32.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/$O/prof_f32 -name "*results.db" | head -1) $R/$O/prof_f32_stats.md; head -24 $R/$O/prof_f32_stats.md | cut -c1-160
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_f32_0 -o r -- python $R/bench.py --dtype f32 --fuse 0 --steps 1 --warmup 1 --streams 1 --no-rtf --no-cpu-baseline > $R/$O/prof_f32_0.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/$O/prof_f32_0 -name "*results.db" | head -1) $R/$O/prof_f32_0_stats.md; head -20 $R/$O/prof_f32_0_stats.md | cut -c1-160
find $R/$O -name "*.db" -size +20M -delete
