#!/bin/bash
# round 3, GPU call 1: co-issue microbenchmark, pair-kernel variants (paced MFMA loops, residual on the matrix cores), GPU tests
O=gpurun_out/r03_exp1; mkdir -p $O
timeout 120 tools/kbench/bin/coissue > $O/coissue.txt 2>&1; echo "coissue rc=$?"
run() { # tag C K dil B L
  for v in base res nop1 nop3 nop5 nop7 nop11; do
    echo "== $v $*" >> $O/kbench.txt
    timeout 120 tools/kbench/bin/kbench_$v $* 7 >> $O/kbench.txt 2>&1
  done
}
run 128 11 3 64 65536
run 128 3 1 64 65536
run 64 11 3 64 131072
run 32 11 3 64 262144
run 32 7 5 64 262144
run 256 11 5 64 8192
for v in base res; do for sh in "128 11 5 64 65536" "64 7 5 64 131072" "32 7 5 64 262144" "256 7 5 64 8192"; do
  echo "== $v ACC $sh" >> $O/kbench.txt; KB_ACC=1 timeout 120 tools/kbench/bin/kbench_$v $sh 7 >> $O/kbench.txt 2>&1; done; done
grep -a "==\|time:\|check:" $O/kbench.txt | paste - - - | sed -e 's/time: min//' | cut -c1-200
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --no-f32 > $O/bench_quick.json 2> $O/bench_quick.err; cut -c1-300 $O/bench_quick.json
