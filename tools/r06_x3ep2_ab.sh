#!/bin/bash
# split-operand engine (round 6 experiment, NOT in the tree any more: measured 2 % slower, profiles/r06_e_x3_pair_epilogue2_staged_ab.txt): epilogue 2 of the pair
# kernel through an LDS transposition area (libvtts_hifigan.so of that build) against the accumulator-layout dword form (libvtts_ep2direct.so =
# --define VTTS_X3_EP2_STAGED=0): tests, interleaved pass times, per-launch durations (rocprofv3)
T=${1:-r06_x3ep2}; R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_x3.py tests/test_gpu_ragged_f32.py tests/test_gpu_longform.py -m gpu -q -x --timeout 600 2>&1 | tail -2
for rep in 1 2 3; do for v in libvtts_hifigan.so libvtts_ep2direct.so; do
  VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v python bench.py --dtype bf16x3 --no-cpu-baseline --no-f32 --no-rtf --steps 5 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'ms_per_step', round(d['ms_per_step'],3))"
done; done | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in libvtts_hifigan.so libvtts_ep2direct.so; do
  VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_$v -o r -- python $R/bench.py --dtype bf16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 --streams 1 --microbatch 64 > $O/$v.log 2>&1
  python $R/tools/rocprof_summary.py $(find $O/p_$v -name "*results.db" | head -1) $O/$v.md; echo "== $v"; grep -E "resblock_pair_x3_k" $O/$v.md | cut -c1-130; rm -rf $O/p_$v
done | tee -a $O/ab.txt
