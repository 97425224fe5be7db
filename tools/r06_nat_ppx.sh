#!/bin/bash
# NAT decoder development: per-launch durations of the decoder kernels for timing-switch builds (--define VTTS_NAT_PP_EXP=1 --define VTTS_PPX=<1,2,3> --libname libvtts_ppx<N>.so: results wrong)
T=${1:-r06_ppx}; shift; R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v VTTS_NAT_PP_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_$v -o r -- python $R/tools/pipeline_bench.py 256 2 x3 > $O/$v.log 2>&1
  python $R/tools/rocprof_summary.py $(find $O/p_$v -name "*results.db" | head -1) $O/$v.md; echo "== $v"; grep -E "nat_dec_p" $O/$v.md | cut -c1-60,150-190; rm -rf $O/p_$v
done | tee $O/ppx.txt
