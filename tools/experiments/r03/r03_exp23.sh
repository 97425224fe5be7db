#!/bin/bash
O=gpurun_out/r03_exp23; mkdir -p $O; R=$PWD
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_longform.py -m gpu -q -x --timeout 600 2>&1 | tail -2
bash tools/ab_bench.sh 3 libvtts_prev.so libvtts_hifigan.so > $O/ab.txt 2>&1; cat $O/ab.txt
for v in libvtts_prev.so libvtts_hifigan.so; do
(cd /tmp && export TMPDIR=/tmp && VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_$v -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 > $R/$O/trace_$v.log 2>&1)
python tools/rocprof_summary.py $(find $O/trace_$v -name "*results.db" | head -1) $O/stats_$v.md; echo "== $v"; grep "resblock_bf16_k" $O/stats_$v.md | cut -c1-120
done
find $O -name "*.db" -size +20M -delete
