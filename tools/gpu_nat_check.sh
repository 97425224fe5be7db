#!/bin/bash
# NAT + pipeline check: GPU tests of the NAT models and the ragged generator batch, the 256-sentence stage timer, rocprof kernel stats of it
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_nat.py tests/test_gpu_bf16.py -m gpu -q -x --timeout 600 -k "nat or ragged or golden or fused_resblock_equals" 2>&1 | tail -8
timeout 300 python tools/pipeline_bench.py 256 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp; R=/root/repo
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_pipe -o pp -- python $R/tools/pipeline_bench.py 256 > $R/gpurun_out/prof_pipe.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_pipe -name "*results.db" | head -1) $R/gpurun_out/prof_pipe_stats.md; head -14 $R/gpurun_out/prof_pipe_stats.md
