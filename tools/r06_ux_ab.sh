#!/bin/bash
# split-operand engine: the stride-2 upsamplers' LDS-staged epilogue (default) against the direct float2 stores (libvtts_uxdirect.so = --define VTTS_UX_STAGED=0):
# tests, per-launch durations (rocprofv3) and the pass time, interleaved
T=${1:-r06_ux}; R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_ragged_f32.py -m gpu -q -x --timeout 600 2>&1 | tail -2
for rep in 1 2 3; do for v in libvtts_hifigan.so libvtts_uxdirect.so; do
  VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v python bench.py --dtype bf16x3 --no-cpu-baseline --no-f32 --no-rtf --steps 5 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'ms_per_step', round(d['ms_per_step'],3))"
done; done | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in libvtts_hifigan.so libvtts_uxdirect.so; do
  VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_$v -o r -- python $R/bench.py --dtype bf16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 --streams 1 --microbatch 64 > $O/$v.log 2>&1
  python $R/tools/rocprof_summary.py $(find $O/p_$v -name "*results.db" | head -1) $O/$v.md; echo "== $v"; grep -E "convt_x3_k" $O/$v.md | cut -c1-130; rm -rf $O/p_$v
done | tee -a $O/ab.txt
