#!/bin/bash
# where the C = 32 kernels' cycles go, in counters: the stage kernel (option "stage" = 1) and the launches it replaces (= 0), one --pmc pass each
#   VALU share = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (quad-cycles a wave spends issuing VALU), MfmaUtil as tools/profile_digest.py
T=${1:-r06_stage_pmc}; R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for st in 1 0; do
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/p$st -- python $R/bench.py --stage $st --steps 1 --warmup 1 --streams 1 --microbatch 64 --no-cpu-baseline --no-rtf --no-f32 > $O/p$st.log 2>&1
done
python - $O <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
for st in (1, 0):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{sys.argv[1]}/p{st}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== option stage = {st}:  kernel | launches | MfmaUtil | VALU issue / wave cycles | wait_any / wave cycles | wait_inst / wave cycles | LDS issue / wave cycles | VALU instructions per launch")
    for k, c in rows.items():
        if not any(s in k for s in ("stage_bf16_k", "RBTile<32", "GTile<32", "GTail")): continue
        n = len(c["GRBM_GUI_ACTIVE"]); m = lambda name: sum(c[name]) / max(len(c[name]), 1)
        util = sum(a / (g / 8 * 1024) for a, g in zip(c["SQ_VALU_MFMA_BUSY_CYCLES"], c["GRBM_GUI_ACTIVE"])) / n
        wc = m("SQ_WAVE_CYCLES")
        print(f"{k.replace('void vtts::','')[:62]:62s} | {n} | {util:.3f} | {m('SQ_ACTIVE_INST_VALU')/wc:.3f} | {m('SQ_WAIT_ANY')/wc:.3f} | {m('SQ_WAIT_INST_ANY')/wc:.3f} | {m('SQ_ACTIVE_INST_LDS')/wc:.3f} | {m('SQ_INSTS_VALU'):.3e}")
PY
find $O -name "*.csv" -size +5M -delete
