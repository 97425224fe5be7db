#!/bin/bash
# Counters for the build you bench.  On one MI355X box:
#     gpurun --timeout 1500 -- 'bash tools/profile_final.sh r04_final'
# runs, for THE SAME bench command (bf16, B = 64 x T = 1024):
#   1. rocprofv3 --kernel-trace --stats                      -> per-kernel durations
#   2. rocprofv3 --kernel-trace --pmc <SQ set + GRBM>        -> MfmaUtil, wait shares, LDS bank conflicts
#   3. rocprofv3 --kernel-trace --pmc FETCH_SIZE             -> HBM bytes read   (own pass: TCC slots, MI355X_MICROARCH.md)
#   4. rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
# (counter passes never combined with sys/hip/hsa tracing) and digests them with tools/profile_digest.py into
#   gpurun_out/<tag>/<tag>_kernel_stats.md, <tag>_pmc.md, counters_bf16.json
# Copy those three into profiles/ (counters_bf16.json keeps its name): bench.py reports roofline.traffic / roofline.mfma_util from
# counters_bf16.json only while its `source_digest` equals the digest of the sources the loaded library was built from.
R=$PWD; T=${1:-r04_final}; O=$R/gpurun_out/$T; mkdir -p $O
# one stream, one pass of 64: a kernel's duration and counters are its own (the engine's default runs two half-size passes side by side;
# bench.py's roofline calibration pass times this same schedule)
BENCH="python $R/bench.py --steps 2 --warmup 1 --streams 1 --microbatch 64 --no-cpu-baseline --no-rtf --no-f32"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- $BENCH > $O/trace.log 2>&1
i=0
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc/p$i -- $BENCH > $O/pmc_p$i.log 2>&1
done
# the fp32 (1e-4 parity) engine at the same shape: durations + the SQ set (its kernels write exactly the algorithmic bytes: profiles/r01_b_pmc_f32.md)
BENCH32="python $R/bench.py --dtype f32 --streams 1 --microbatch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-rtf"  # one stream: a kernel's duration and counters are its own (the engine's default runs two half-size passes side by side)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_f32 -o r -- $BENCH32 > $O/trace_f32.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_f32/p1 -- $BENCH32 > $O/pmc_f32_p1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f32/p2 -- $BENCH32 > $O/pmc_f32_p2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_f32/p3 -- $BENCH32 > $O/pmc_f32_p3.log 2>&1
# the split-operand engine (VTTS_BF16X3) at the same shape, same schedule
BENCHX3="python $R/bench.py --dtype bf16x3 --streams 1 --microbatch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32"  # --no-f32: without it the fp32 side leg ran inside this trace and its 0.85-MfmaUtil kernels were averaged into the x3 figure (round 4: 0.691 committed, 0.50 true)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_bf16x3 -o r -- $BENCHX3 > $O/trace_bf16x3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_bf16x3/p1 -- $BENCHX3 > $O/pmc_bf16x3_p1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_bf16x3/p2 -- $BENCHX3 > $O/pmc_bf16x3_p2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_bf16x3/p3 -- $BENCHX3 > $O/pmc_bf16x3_p3.log 2>&1
cd $R
python tools/profile_digest.py $O $T
python tools/profile_digest.py $O $T f32
python tools/profile_digest.py $O $T bf16x3
find $O -name "*.csv" -size +5M -delete; find $O -name "*.db" -size +20M -delete
