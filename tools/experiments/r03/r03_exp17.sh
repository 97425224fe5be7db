#!/bin/bash
O=gpurun_out/r03_exp17; mkdir -p $O; R=$PWD
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 > $R/$O/trace.log 2>&1)
python tools/rocprof_summary.py $(find $O/trace -name "*results.db" | head -1) $O/stats.md; grep "convt_g\|conv_bf16_k\|conv_post" $O/stats.md | cut -c1-150
bash tools/ab_bench.sh 2 libvtts_ug0.so libvtts_hifigan.so > $O/ab.txt 2>&1; cat $O/ab.txt
find $O -name "*.db" -size +20M -delete
