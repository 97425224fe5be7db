#!/usr/bin/env python3
"""Per-region instruction counts of a kernel's ISA (tools/kbench/cc_one.sh writes /tmp/<stem>.s): regions end at barriers, branches and labels.
    python tools/kbench/isa_regions.py /tmp/kernels_bf16_stage.s [min_instructions]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
minv = int(sys.argv[2]) if len(sys.argv) > 2 else 20
K = ('mfma', 'valu', 'ds', 'vmem', 'sst', 'sld')
cur = dict(start=0, **{k: 0 for k in K}); reg = []
def flush(i, why):
    global cur
    cur['end'] = i; cur['why'] = why; reg.append(cur); cur = dict(start=i, **{k: 0 for k in K})
for i, l in enumerate(lines):
    t = l.strip()
    if t.startswith('s_barrier'): flush(i, 'barrier')
    elif t.startswith('s_cbranch') or t.startswith('s_branch'): flush(i, ' '.join(t.split()[:2]))
    elif re.match(r'^\.LBB\d+_\d+:', t): flush(i, t.split()[0])
    elif t.startswith('v_mfma'): cur['mfma'] += 1
    elif t.startswith('scratch_store'): cur['sst'] += 1
    elif t.startswith('scratch_load'): cur['sld'] += 1
    elif t.startswith('ds_'): cur['ds'] += 1
    elif t.startswith('global_') or t.startswith('buffer_'): cur['vmem'] += 1
    elif t.startswith('v_'): cur['valu'] += 1
for r in reg:
    if r['mfma'] or r['sst'] or r['sld'] or r['valu'] > minv:
        print(f"{r['start']:6d}-{r['end']:6d} {r['why'][:28]:28s} " + ' '.join(f"{k} {r[k]:4d}" for k in K))
