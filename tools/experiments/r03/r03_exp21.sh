#!/bin/bash
O=gpurun_out/r03_exp21; mkdir -p $O; R=$PWD
for v in libvtts_prev.so libvtts_hifigan.so; do
(cd /tmp && export TMPDIR=/tmp && VTTS_HIFIGAN_LIB=$R/viettts_amd/lib/$v timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_$v -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rtf --no-f32 > $R/$O/trace_$v.log 2>&1)
python - <<PY
import sqlite3,glob
f=glob.glob('$O/trace_$v/**/*results.db',recursive=True)[0]
db=sqlite3.connect(f)
rows=db.execute("select name,start,end from kernels order by start").fetchall()
from collections import defaultdict
d=defaultdict(list)
for n,s,e in rows:
    if 'GTile<32' in n or 'conv_post' in n: d[n.split('(')[0][-70:]].append((e-s)/1e3)
print('== $v')
for k,v in d.items():
    n=len(v); t=v[n//3:]
    print(k[-62:], ' '.join('%.0f'%x for x in t))
PY
done
find $O -name "*.db" -size +20M -delete
