#!/bin/bash
# runs on the GPU box: every library variant (built on the CPU box by pkfma_build.sh) through pkfma_bisect.py
R=$PWD
for lib in ${LIBS:-libvtts_hifigan.so libvtts_slpnat.so libvtts_pk7.so libvtts_pk1.so libvtts_pk2.so libvtts_pk4.so}; do
  for beside in ${BESIDES:-bf16}; do
    echo "=== $lib beside=$beside"
    VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$lib timeout 300 python tools/experiments/r05/pkfma_bisect.py $beside ${REPS:-4} 2>&1 | grep -E "rep |RESULT|Error|error" | cut -c1-260
  done
done
