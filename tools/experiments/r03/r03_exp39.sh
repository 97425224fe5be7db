#!/bin/bash
# how much of the weight-fragment stream do the vector L1s absorb?  TCP (L1) accesses vs TCP -> TCC (L2) read requests per kernel
O=gpurun_out/r03_exp39; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCP_[A-Z0-9_a-z]*\|TCC_REQ[A-Z0-9_a-z]*\|TCC_READ[A-Z0-9_a-z]*\|TCC_EA0_RDREQ[A-Z0-9_a-z]*" | sort -u > $R/$O/avail.txt
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32"
timeout 600 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $R/$O/p1 -- $BENCH > $R/$O/p1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum --output-format csv -d $R/$O/p2 -- $BENCH > $R/$O/p2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for p in ("p1", "p2"):
    f = glob.glob(f"gpurun_out/r03_exp39/{p}/**/*counter_collection.csv", recursive=True)
    if not f: print(p, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"][:70]; acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for row in csv.DictReader(open(f[0])):
        pass
    disp = collections.defaultdict(set)
    for row in csv.DictReader(open(f[0])): disp[row["Kernel_Name"][:70]].add(row["Dispatch_Id"])
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:14]:
        nd = len(disp[k]); print(p, k, {c: "%.3e" % (x / nd) for c, x in v.items()}, "launches", nd)
PY
find $O -name "*.csv" -size +5M -delete
