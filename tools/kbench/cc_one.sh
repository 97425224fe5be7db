#!/bin/bash
# compile ONE kernel file of the library the way build.py does and print its resource usage (kernel-development helper)
#   tools/kbench/cc_one.sh kernels_bf16_stage.hip [extra flags]      -> /tmp/<stem>.o, /tmp/<stem>.s
R=$(cd "$(dirname "$0")/../.." && pwd); f=$1; shift; st=$(basename $f .hip)
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -fno-slp-vectorize"
/opt/rocm/bin/hipcc $F "$@" -Rpass-analysis=kernel-resource-usage -c $R/viettts_amd/csrc/$f -o /tmp/$st.o 2>&1 | grep -E "error|Function Name|VGPRs|Scratch|Occupancy|warning" | sed 's/.*remark: *//'
/opt/rocm/bin/hipcc $F "$@" -S --cuda-device-only $R/viettts_amd/csrc/$f -o /tmp/$st.s 2>/dev/null
echo "scratch_store $(grep -c scratch_store /tmp/$st.s) scratch_load $(grep -c scratch_load /tmp/$st.s) mfma $(grep -c v_mfma /tmp/$st.s)"
