#!/usr/bin/env python3
"""VGPR liveness of a straight-line stretch of a kernel's ISA (tools/kbench/cc_one.sh writes /tmp/<stem>.s): backward liveness over lines [a, b) taken as a
loop body (two passes, so loop-carried values count), printing the live-register count every `step` lines and the maximum.
    python tools/kbench/isa_pressure.py /tmp/kernels_bf16_stage.s 7142 9612 [step]"""
import re, sys
L = open(sys.argv[1]).read().split('\n')
a, b = int(sys.argv[2]), int(sys.argv[3])
step = int(sys.argv[4]) if len(sys.argv) > 4 else 100
def vregs(op):
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', op)
    if m: return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', op)
    return [int(m.group(1))] if m else []
ins = []
for i in range(a, b):
    t = L[i].split(';')[0].strip()
    if not t or t.endswith(':') or t.startswith('.') or t.startswith('s_') or t.startswith(';'):
        continue
    parts = t.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in re.split(r',\s*(?![^\[]*\])', parts[1])] if len(parts) > 1 else []
    ops = [o.split()[0] if o else o for o in ops]
    if 'store' in op or op.startswith('ds_write'):
        d, u = [], sum((vregs(o) for o in ops), [])
    elif op.startswith('v_permlane32_swap') or op.startswith('v_swap'):
        d = u = sum((vregs(o) for o in ops), [])
    else:
        d = vregs(ops[0]) if ops else []
        u = sum((vregs(o) for o in ops[1:]), [])
        if op.startswith('v_mac') or op.startswith('v_fmac') or op.startswith('v_dot2c'): u = u + d
    ins.append((i, d, u))
live = set()
for _ in range(2):
    rec = []
    for i, d, u in reversed(ins):
        live -= set(d)
        live |= set(u)
        rec.append((i, len(live)))
rec.reverse()
mx = max(rec, key=lambda r: r[1])
for i, n in rec[::step]: print(i, n)
print('max', mx)
