#!/bin/bash
# re-measure round 2's persistent software-pipelined pair kernel (tools/kbench/experiments/kernels_bf16_rbp.hip) against the product pair kernel with
# round 3's / 4's build rules (no packed-f32 VALU: lrelu01_pack is scalar now, -fno-slp-vectorize): C = 128, k = 11, the three rates
O=gpurun_out/r04_run14; mkdir -p $O
for d in 1 3 5; do for impl in 0 1 0 1; do
  echo "== dil $d impl $impl"; timeout 120 tools/kbench/bin/kbench_p 128 11 $d 64 65536 5 $impl 2>&1 | tail -6
done; done | tee $O/kbench_p.log | grep -a "==\|ms\|err\|TF" | cut -c1-200
