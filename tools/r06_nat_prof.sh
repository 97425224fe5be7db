#!/bin/bash
# NAT decoder development (needs an experiment build: python -m viettts_amd.csrc.build --define VTTS_NAT_PP_EXP=1 --libname libvtts_ppexp.so, then VTTS_HIFIGAN_LIB=...): tests, a short A/B of option "pp_split", and the decoder kernels' per-launch durations (rocprofv3) with the option on
T=${1:-r06_nat}; R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nat.py -m gpu -q -x --timeout 600 2>&1 | tail -3
for rep in 1 2; do for pp in 1 0; do
  VTTS_NAT_PP_SPLIT=$pp python tools/pipeline_bench.py 256 3 x3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pp_split', $pp, 'x3 acoustic_model_ms', round(d['acoustic_model_ms'],2), 'total_ms', round(d['total_ms'],2))"
done; done | tee $O/ab_short.txt
cd /tmp && export TMPDIR=/tmp
VTTS_NAT_PP_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/p -o r -- python $R/tools/pipeline_bench.py 256 3 x3 > $O/prof.log 2>&1
python $R/tools/rocprof_summary.py $(find $O/p -name "*results.db" | head -1) $O/stats.md; grep -E "nat_dec" $O/stats.md | cut -c1-150; rm -rf $O/p
