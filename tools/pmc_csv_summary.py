#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean counter value per dispatch.
Usage: tools/pmc_csv_summary.py <dir-with-csv> [name-filter]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt and flt not in k:
            continue
        acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
