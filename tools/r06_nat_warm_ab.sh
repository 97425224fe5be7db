#!/bin/bash
# NAT decoder (round 6 experiment, NOT in the tree any more: neutral-to-worse, profiles/r06_c_nat_warm_ab.txt): nat_dec_proj_prenet_k with its idle wave touching every line of
# the prenet matrices at the kernel's start (libvtts_hifigan.so of that build) against without (libvtts_nowarm.so = --define VTTS_NAT_PP_WARM=0): tests, interleaved per-stage
# pipeline times, per-launch durations


T=${1:-r06_warm}; R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nat.py -m gpu -q -x --timeout 600 -k "acoustic or text2mel" 2>&1 | tail -2
for rep in 1 2 3; do for v in libvtts_hifigan.so libvtts_nowarm.so; do for mode in x3 fp32; do
  VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v python tools/pipeline_bench.py 256 3 $mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$mode', 'acoustic_model_ms', round(d['acoustic_model_ms'],2), 'total_ms', round(d['total_ms'],2))"
done; done; done | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in libvtts_hifigan.so libvtts_nowarm.so; do
  VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_$v -o r -- python $R/tools/pipeline_bench.py 256 2 x3 > $O/$v.log 2>&1
  python $R/tools/rocprof_summary.py $(find $O/p_$v -name "*results.db" | head -1) $O/$v.md; echo "== $v"; grep -E "nat_dec" $O/$v.md | cut -c1-60,140-190; rm -rf $O/p_$v
done | tee -a $O/ab.txt
