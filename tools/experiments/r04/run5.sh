#!/bin/bash
O=gpurun_out/r04_run5; mkdir -p $O
echo "== product build (packed-f32 VALU in the NAT kernels)"; timeout 600 python tools/experiments/r04/diag_pipe3.py short 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee $O/diag_base.log
echo "== nat.hip built with -fno-slp-vectorize (no v_pk_*_f32)"; VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/libvtts_natnoslp.so timeout 600 python tools/experiments/r04/diag_pipe3.py short 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee $O/diag_noslp.log
