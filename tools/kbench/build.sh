#!/bin/bash
# build the kernel-development harness (gfx950 cross-compile; the binary travels to the GPU box with the tree)
set -e
cd "$(dirname "$0")/../.."
OUT=${1:-tools/kbench/kbench}; shift || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -I viettts_amd/csrc "$@" \
  -DKBENCH_NO_P tools/kbench/kbench.hip viettts_amd/csrc/kernels_bf16.hip viettts_amd/csrc/kernels_bf16_rbg.hip -o "$OUT"
echo "built $OUT"
