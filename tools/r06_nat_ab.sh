#!/bin/bash
# NAT decoder: the projection + prenet step cut along its weights (option "pp_split") on / off, interleaved three times on one box: per-stage wall times of the
# 256-sentence pipeline (tools/pipeline_bench.py), both acoustic modes.  Needs an experiment build of the library:
#   python -m viettts_amd.csrc.build --define VTTS_NAT_PP_EXP=1 --libname libvtts_ppexp.so;  gpurun -- 'VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/libvtts_ppexp.so bash tools/r06_nat_ab.sh [tag]'
T=${1:-r06_nat}; R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O
for rep in 1 2 3; do for pp in 1 0; do for mode in x3 fp32; do
  VTTS_NAT_PP_SPLIT=$pp python tools/pipeline_bench.py 256 3 $mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pp_split', $pp, '$mode', 'acoustic_model_ms', round(d['acoustic_model_ms'],2), 'total_ms', round(d['total_ms'],2), 'generator_ms', round(d['generator_ms'],2))"
done; done; done | tee $O/ab.txt
