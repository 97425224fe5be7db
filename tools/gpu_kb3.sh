R=$PWD; cd tools/kbench
for cfg in "128 3 1 16 65536" "128 11 3 16 65536"; do
 for impl in 1 2; do
  for fl in 0 1 2 4 8 16 32 7 15 48; do
    echo "== cfg $cfg impl $impl flags $fl: $(timeout 60 ./kbench_tl $cfg 3 $impl $fl | grep -E '^time' )"
  done
  timeout 60 ./kbench_tl $cfg 3 $impl 0 | grep -A12 timeline
 done
done
cd /tmp && export TMPDIR=/tmp
for impl in 0 1 2; do
for cfg in "128 3 1 16 65536" "128 11 3 16 65536"; do
 tag=$(echo $cfg | tr ' ' '_')_i$impl
 timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_kb3/sq_$tag -- $R/tools/kbench/kbench $cfg 2 $impl > /dev/null 2>&1
 timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_kb3/fe_$tag -- $R/tools/kbench/kbench $cfg 2 $impl > /dev/null 2>&1
 timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_kb3/wr_$tag -- $R/tools/kbench/kbench $cfg 2 $impl > /dev/null 2>&1
 echo "#### PMC $tag"; python $R/tools/pmc_csv_summary.py $R/gpurun_out/pmc_kb3 resblock 2>&1 | tail -40
 rm -rf $R/gpurun_out/pmc_kb3
done
done
