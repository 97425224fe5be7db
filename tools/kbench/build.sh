#!/bin/bash
# build the kernel-development harness (gfx950 cross-compile; the binary travels to the GPU box with the tree)
set -e
cd "$(dirname "$0")/../.."
# KBENCH_P=1: also link the persistent software-pipelined experiment (tools/kbench/experiments/kernels_bf16_rbp.hip, impl = 1;
# add -DVTTS_P_ONLY_K11_D3 to compile one instantiation only, -DVTTS_TIMELINE=1 -DPEXP=<bits> for its ablations)
OUT=${1:-tools/kbench/kbench}; shift || true
if [ -n "$KBENCH_P" ]; then EXTRA="tools/kbench/experiments/kernels_bf16_rbp.hip"; else EXTRA="-DKBENCH_NO_P"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-function -I viettts_amd/csrc "$@" \
  $EXTRA tools/kbench/kbench.hip viettts_amd/csrc/kernels_bf16.hip viettts_amd/csrc/kernels_bf16_rbg.hip -o "$OUT"
echo "built $OUT"
