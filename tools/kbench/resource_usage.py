#!/usr/bin/env python3
"""VGPRs / spills / occupancy per kernel from `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr file): tools/kbench/resource_usage.py FILE [filter]"""
import re
import subprocess
import sys

cur = None
rows = {}
for line in open(sys.argv[1]):
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in rows.items():
    if flt in k:
        print(f"{k[:110]:110s} VGPR {v.get('VGPRs'):4d} AGPR {v.get('AGPRs'):3d} scratch {v.get('ScratchSize'):4d} spill {v.get('VGPRs Spill'):3d} occ {v.get('Occupancy')}")
