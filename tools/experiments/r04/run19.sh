#!/bin/bash
# LSTM step: iterations in flight (PD 4 / 8 / 12); and is the L2 cold at every launch because of capacity or because of the kernel boundary? (64 sentences: a quarter of the state)
O=gpurun_out/r04_run19; mkdir -p $O
for rep in 1 2; do for v in libvtts_hifigan.so libvtts_pd8.so libvtts_pd12.so; do
echo -n "$v  "; VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/$v timeout 300 python tools/pipeline_bench.py 256 1 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('acoustic_model_ms','total_ms')})"; done; done
cd /tmp && export TMPDIR=/tmp; R=/root/repo
for n in 64 256; do
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$O/p3_$n -- python $R/tools/pipeline_bench.py $n 1 2 > $R/$O/p3_$n.log 2>&1
echo "== TCC, $n sentences"; python $R/tools/pmc_csv_summary.py $R/$O/p3_$n nat_dec | cut -c1-120
done
for v in libvtts_pd8.so; do
VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$v -o r -- python $R/tools/pipeline_bench.py 256 1 3 > $R/$O/prof_$v.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/$O/prof_$v -name "*results.db" | head -1) $R/$O/prof_$v.md; grep "nat_dec" $R/$O/prof_$v.md | cut -c1-150
done
find $R/$O -name "*.db" -delete; find $R/$O -name "*.csv" -size +5M -delete
