#!/bin/bash
O=gpurun_out/r03_exp16; mkdir -p $O; R=$PWD
timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x --timeout 600 2>&1 | tail -2
for v in libvtts_ug0.so libvtts_nostage.so libvtts_hifigan.so; do
  (cd /tmp && export TMPDIR=/tmp && VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_$v -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 > $R/$O/trace_$v.log 2>&1)
  python tools/rocprof_summary.py $(find $O/trace_$v -name "*results.db" | head -1) $O/stats_$v.md; echo "== $v"; grep "convt_g\|conv_bf16_k" $O/stats_$v.md | cut -c1-140
done
bash tools/ab_bench.sh 3 libvtts_ug0.so libvtts_hifigan.so > $O/ab.txt 2>&1; cat $O/ab.txt
for v in libvtts_ug0.so libvtts_hifigan.so; do echo -n "$v "; VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/$v timeout 300 python tools/pipeline_bench.py 256 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gen %.2f total %.2f'%(d['generator_ms'],d['total_ms']))"; done
find $O -name "*.db" -size +20M -delete
