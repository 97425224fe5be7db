"""option "zigzag" (consecutive launches walk the batch in alternating directions) 0 vs 1, both engines, 64 x 1024 frames, interleaved"""
import sys, time
import numpy as np
import torch
from viettts_amd.hifigan.config import V1
from viettts_amd.hifigan.generator import Generator
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params

dev = torch.device("cuda:0")
params = synthetic_params(V1, 4321, "scaled")
mel = torch.from_numpy(synthetic_mel(64, 1024, 1234)).to(dev)
out = torch.empty((64, 256 * 1024), device=dev)
for dtype, steps in (("bf16", 8), ("f32", 2)):
    g = Generator(V1, device=dev, dtype=dtype)
    g.load_params(params)
    res, outs = {}, {}
    for rep in range(3):
        for z in (0, 1):
            g.set_option("zigzag", z)
            g(mel, out)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(steps): g(mel, out)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t) / steps * 1e3
            res.setdefault(z, []).append(ms)
            outs[z] = out.clone() if rep == 0 else outs[z]
    print(dtype, {z: [round(v, 2) for v in vs] for z, vs in res.items()}, "same bits:", bool(torch.equal(outs[0], outs[1])), flush=True)
    g.close()
