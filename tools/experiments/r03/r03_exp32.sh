#!/bin/bash
# power / clock of the GPU under (a) the bf16 bench pass, (b) an MFMA-only stream on random / constant / zero operands, (c) the fp32 pass
O=gpurun_out/r03_exp32; mkdir -p $O
ls /sys/class/drm/card*/device/hwmon/hwmon*/ > $O/hwmon_ls.txt 2>&1
rocm-smi --showpower --showclocks --showmaxpower --json > $O/idle_smi.json 2>&1
timeout 200 python tools/power_trace.py $O/bench_bf16.json -- python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-rtf --no-f32 > $O/log.txt 2>&1
for dm in 2 1 0; do timeout 60 python tools/power_trace.py $O/mfma_only_dm$dm.json -- tools/kbench/bin/coissue 400 4 $dm 8 >> $O/log.txt 2>&1; done
timeout 200 python tools/power_trace.py $O/bench_f32.json -- python bench.py --dtype f32 --steps 20 --warmup 2 --no-cpu-baseline --no-rtf >> $O/log.txt 2>&1
cat $O/log.txt
