#!/bin/bash
# the stage kernel in the latency regime: batch-1 latency (B = 1, T = 512; and T = 128, 2048) with option "stage" on / off, interleaved
python - <<'PY'
import time, statistics, torch, sys
sys.path.insert(0, '.')
from viettts_amd.hifigan.config import V1
from viettts_amd.hifigan.generator import Generator
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params
g = Generator(V1, device="cuda:0", dtype="bf16")
g.load_params(synthetic_params(V1, 4321, "scaled"))
for B, T in ((1, 128), (1, 512), (1, 2048), (4, 512), (8, 1024)):
    m = torch.from_numpy(synthetic_mel(B, T, 1234)).to("cuda:0")
    o = torch.empty((B, 256 * T), dtype=torch.float32, device="cuda:0")
    res = {}
    for rep in range(3):
        for st in (1, 0):
            g.set_option("stage", st)
            for _ in range(12): g(m, o)
            torch.cuda.synchronize()
            lat = []
            for _ in range(30):
                t = time.perf_counter(); g(m, o); torch.cuda.synchronize(); lat.append(time.perf_counter() - t)
            res.setdefault(st, []).append(statistics.median(lat) * 1e3)
    print(f"B={B} T={T}: stage=1 {[round(v,4) for v in res[1]]} ms   stage=0 {[round(v,4) for v in res[0]]} ms")
PY
