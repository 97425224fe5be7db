#!/bin/bash
# round 4, GPU call 2: pipeline diagnostic, fused fp32 pairs (tests + A/B over fuse levels), kernel traces of the pipeline
O=gpurun_out/r04_run2; mkdir -p $O
timeout 600 python tools/experiments/r04/diag_pipe.py 12 > $O/diag.log 2>&1; echo "diag rc=$?"; cat $O/diag.log | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "fp32_fused or golden or edge or baseline_config2" > $O/pytest_f32.log 2>&1; echo "f32 rc=$?"; tail -5 $O/pytest_f32.log
for f in 0 2 3 0 2 3; do
  echo -n "fuse=$f " >> $O/f32_ab.log
  timeout 300 python bench.py --dtype f32 --fuse $f --steps 2 --warmup 1 --no-rtf --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.2f frac %.4f roof %.4f'%(d['ms_per_step'], d['frac_of_mfma_peak_whole_forward'], d['roofline']['frac']))" >> $O/f32_ab.log
done
cat $O/f32_ab.log
cd /tmp && export TMPDIR=/tmp; R=/root/repo
for og in 1 4; do
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_pipe$og -o r -- python $R/tools/pipeline_bench.py 256 $og 3 > $R/$O/prof_pipe$og.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/$O/prof_pipe$og -name "*results.db" | head -1) $R/$O/prof_pipe${og}_stats.md; head -30 $R/$O/prof_pipe${og}_stats.md | cut -c1-200
done
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_f32 -o r -- python $R/bench.py --dtype f32 --fuse 3 --steps 1 --warmup 1 --streams 1 --no-rtf --no-cpu-baseline > $R/$O/prof_f32.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/$O/prof_f32 -name "*results.db" | head -1) $R/$O/prof_f32_stats.md; head -34 $R/$O/prof_f32_stats.md | cut -c1-200
find $R/$O -name "*.db" -size +20M -delete
