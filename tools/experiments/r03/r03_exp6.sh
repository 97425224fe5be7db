#!/bin/bash
O=gpurun_out/r03_exp6; mkdir -p $O; R=$PWD
# fp32 engine: SLP on/off in kernels_f32_mfma.hip
for r in 1 2 3; do for v in libvtts_hifigan.so libvtts_f32noslp.so; do echo -n "$v " ; VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/$v timeout 300 python bench.py --dtype f32 --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-rtf 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f32 value %.4e ms %.2f roof %.3f'%(d['value'],d['ms_per_step'],d['roofline']['frac']))"; done; done > $O/ab_f32.txt 2>&1; cat $O/ab_f32.txt
# NAT: SLP on/off in nat.hip (pipeline stage times)
for r in 1 2 3; do for v in libvtts_hifigan.so libvtts_natnoslp.so; do echo -n "$v "; VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/$v timeout 300 python tools/pipeline_bench.py 256 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dur %.2f ac %.2f gen %.2f total %.2f'%(d['duration_model_ms'],d['acoustic_model_ms'],d['generator_ms'],d['total_ms']))"; done; done > $O/ab_nat.txt 2>&1; cat $O/ab_nat.txt
# whole-ResBlock kernels wherever they exist (fuse 3) vs the default policy (fuse 2)
for r in 1 2; do for f in 2 3; do echo -n "fuse $f "; timeout 300 python bench.py --fuse $f --steps 6 --warmup 2 --no-cpu-baseline --no-f32 --no-rtf 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.4e ms %.2f'%(d['value'],d['ms_per_step']))"; done; done > $O/ab_fuse.txt 2>&1; cat $O/ab_fuse.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_fuse3 -o r -- python $R/bench.py --fuse 3 --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 > $R/$O/trace_fuse3.log 2>&1
cd $R; python tools/rocprof_summary.py $(find $O/trace_fuse3 -name "*results.db" | head -1) $O/fuse3_stats.md; grep "resblock_bf16_k\|GTile<32" $O/fuse3_stats.md | cut -c1-150
bash tools/profile_final.sh r03_b > $O/profile_final.log 2>&1; tail -2 $O/profile_final.log; head -30 gpurun_out/r03_b/r03_b_pmc.md | cut -c1-220
