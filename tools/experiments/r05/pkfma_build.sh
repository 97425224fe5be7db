#!/bin/bash
# CPU box: the library variants of the bisect (next to the product library; they travel with gpurun)
set -e
cd "$(dirname "$0")/../../.."
ALL="engine.hip,kernels_generic.hip,kernels_f32_mfma.hip,kernels_f32_pair.hip,kernels_x3.hip,kernels_x3_rb.hip,kernels_bf16.hip,kernels_bf16_rbg.hip,kernels_bf16_rbk.hip,kernels_bf16_up.hip"
# nat.hip with hipcc's SLP vectoriser (round 4's failing build), everything else as the product
VTTS_BUILD_NO_FILE_FLAGS=1 VTTS_BUILD_NOSLP_FILES=$ALL python -m viettts_amd.csrc.build --libname libvtts_slpnat.so
for bits in ${BITS:-7 1 2 4}; do python -m viettts_amd.csrc.build --libname libvtts_pk$bits.so --define VTTS_NAT_PKFMA=$bits; done
