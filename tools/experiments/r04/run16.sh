#!/bin/bash
# What bounds the NAT decoder's step kernels?  PMC passes of the 256-sentence pipeline (one counter set per pass, no trace domains)
O=gpurun_out/r04_run16; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; R=/root/repo
CMD="python $R/tools/pipeline_bench.py 256 1 2"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
VTTS_NAT_SL=1 VTTS_NAT_AHEAD=0 timeout 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $R/$O/p1 -- $CMD > $R/$O/p1.log 2>&1
VTTS_NAT_SL=1 VTTS_NAT_AHEAD=1 timeout 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $R/$O/p1a -- $CMD > $R/$O/p1a.log 2>&1
VTTS_NAT_SL=1 VTTS_NAT_AHEAD=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/p2 -- $CMD > $R/$O/p2.log 2>&1
VTTS_NAT_SL=1 VTTS_NAT_AHEAD=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$O/p3 -- $CMD > $R/$O/p3.log 2>&1
VTTS_NAT_SL=1 VTTS_NAT_AHEAD=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU --output-format csv -d $R/$O/p4 -- $CMD > $R/$O/p4.log 2>&1
for p in p1 p1a p2 p3 p4; do echo "== $p"; python $R/tools/pmc_csv_summary.py $R/$O/$p nat_dec > $R/$O/$p.txt 2>&1; cat $R/$O/$p.txt | cut -c1-120; done
find $R/$O -name "*.csv" -size +5M -delete; find $R/$O -name "*.db" -delete
