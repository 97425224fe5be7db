# Full GPU check on one MI355X box (gpurun --timeout 2400 -- "bash tools/gpu_full_check.sh [tag]"): pytest -m gpu, smoke, bench.py,
# rocprofv3 --kernel-trace --stats of the bench command and of the pipeline, and the three --pmc passes (SQ / FETCH_SIZE / WRITE_SIZE).
# Outputs under gpurun_out/<tag>*.
R=$PWD; T=${1:-full}; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -a "passed\|failed\|rc=\|bf16 B=" gpurun_out/pytest_gpu.log | tail -5
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err; cut -c1-400 gpurun_out/bench_$T.json; tail -2 gpurun_out/bench_$T.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$T -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 > $R/gpurun_out/prof_$T.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_$T -name "*results.db" | head -1) $R/gpurun_out/prof_${T}_stats.md
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profpipe_$T -o r -- python $R/tools/pipeline_bench.py 256 > $R/gpurun_out/profpipe_$T.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/profpipe_$T -name "*results.db" | head -1) $R/gpurun_out/profpipe_${T}_stats.md
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmc_$T/$tag -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 > $R/gpurun_out/pmc_${T}_$tag.log 2>&1
done
python $R/tools/pmc_csv_summary.py $R/gpurun_out/pmc_$T > $R/gpurun_out/pmc_${T}_summary.txt 2>&1
find $R/gpurun_out -name "*.csv" -size +5M -delete; find $R/gpurun_out -name "*.db" -size +20M -delete
