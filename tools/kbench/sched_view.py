#!/usr/bin/env python3
"""Compact view of a kernel's instruction schedule from hipcc -S output: M = MFMA, r = ds_read_b128, W = ds_write,
G = LDS-DMA, g = global load, S = global store, |B| = s_barrier, w(..) = s_waitcnt.  Usage: sched_view.py file.s name-substring"""
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
for i, l in enumerate(lines):
    if l.startswith("_ZN") and pat in l and ":" in l and not l.startswith("\t"):
        j = i
        while not lines[j].startswith(".Lfunc_end"):
            j += 1
        toks = []
        for b in lines[i:j]:
            b = b.strip()
            t = b.split(" ")[0] if b else ""
            if t.startswith("v_mfma"): toks.append("M")
            elif t.startswith("ds_read"): toks.append("r")
            elif t.startswith("ds_write"): toks.append("W")
            elif t.startswith("global_load_lds"): toks.append("G")
            elif t.startswith("global_load") or t.startswith("buffer_load"): toks.append("g")
            elif t.startswith("global_store"): toks.append("S")
            elif t == "s_barrier": toks.append("|B|")
            elif t == "s_waitcnt": toks.append("w(" + b.split(" ", 1)[1].replace("lgkmcnt", "l").replace("vmcnt", "v").replace(" ", "") + ")")
            elif t.startswith("s_cbranch"): toks.append("^")
            elif t.startswith(".LBB"): toks.append("\n" + t)
        print(l.split(":")[0][-60:], "instructions:", j - i)
        print("".join(toks))
        print()
