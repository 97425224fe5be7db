#!/bin/bash
# MfmaUtil (the counter formula of tools/profile_digest.py) of the isolated k-step loops of tools/kbench/kstep_block.hip: calibrates the counter against a loop
# known to sit on the matrix pipe's floor (profiles/r05_i_kstep_asm_findings.md).  Runs on the GPU box: bash tools/kbench/kstep_pmc.sh [nblk=88]
NBLK=${1:-88}
R=$PWD; O=$R/gpurun_out/r05_kstep_pmc; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $O/p -- $R/tools/kbench/bin/kstep_block $NBLK 5 > $O/run.log 2>&1
python - "$O" <<'PY' | tee $O/../r05_kstep_pmc.txt
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True)[0]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
order = []
for r in csv.DictReader(open(f)):
    key = (r["Kernel_Name"], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""), r.get("LDS_Block_Size", ""))
    if key not in order: order.append(key)
    rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel | grid | LDS/WG | launches | MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024)")
for k in order:
    c = rows[k]
    n = len(c["GRBM_GUI_ACTIVE"])
    u = [m / (g / 8 * 1024) for m, g in zip(c["SQ_VALU_MFMA_BUSY_CYCLES"], c["GRBM_GUI_ACTIVE"]) if g]
    print(f"{k[0][:40]} | {k[1]} | {k[2]} | {n} | mean {sum(u)/len(u):.3f}  max {max(u):.3f}")
PY
find $O -name "*.db" -delete
