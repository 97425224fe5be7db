for q in 2 4 8 16; do echo -n "GPU_MAX_HW_QUEUES=$q "; GPU_MAX_HW_QUEUES=$q python tools/experiments/r04/diag_pipe_bench2.py plain 2>&1 | grep -v amdgpu | tail -1; done
echo -n "GPU_MAX_HW_QUEUES=8 pipefirst "; GPU_MAX_HW_QUEUES=8 python tools/experiments/r04/diag_pipe_bench2.py pipefirst 2>&1 | grep -v amdgpu | tail -1
