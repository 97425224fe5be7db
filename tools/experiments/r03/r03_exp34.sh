#!/bin/bash
# (the engine change this tested was not kept: see profiles/r03_c_narrow_stage_findings.md)
# MRF accumulation order: whole-ResBlock kernels last (option "mrf_order" 1, new default) vs configuration order (0)
O=gpurun_out/r03_exp34; mkdir -p $O; R=$PWD
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x --timeout 600 2>&1 | tail -3
python - > $O/ab.txt 2>&1 <<'PY'
import torch, time, numpy as np
from viettts_amd.hifigan.generator import Generator
from viettts_amd.hifigan.config import V1
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params
dev = torch.device("cuda:0")
g = Generator(V1, device=dev, dtype="bf16"); g.load_params(synthetic_params(V1, 4321, "scaled"))
mel = torch.from_numpy(synthetic_mel(64, 1024, 1234)).to(dev); out = torch.empty((64, 256 * 1024), device=dev)
res = {}
for rep in range(3):
    for mode in (0, 1):
        g.set_option("mrf_order", mode)
        for _ in range(2): g(mel, out)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(6): g(mel, out)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 6 * 1e3
        res.setdefault(mode, []).append(ms)
        print(f"mrf_order {mode}: {ms:.2f} ms", flush=True)
    if rep == 0:
        g.set_option("mrf_order", 0); a = g(mel).clone(); g.set_option("mrf_order", 1); b = g(mel).clone()
        print("max |wav(order 1) - wav(order 0)| = %.3e" % float((a - b).abs().max()))
print({k: round(float(np.mean(v)), 3) for k, v in res.items()})
PY
cat $O/ab.txt
