#!/usr/bin/env python
"""PCIe-inclusive rate of the hot path (never bench.py's `value`): pinned host mel -> HBM -> bf16 generator ->
HBM -> pinned host waveform, B = 64 x T = 1024, (a) one batch at a time, (b) two batches in flight on two torch streams."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viettts_amd.hifigan.config import V1  # noqa: E402
from viettts_amd.hifigan.generator import Generator  # noqa: E402
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params  # noqa: E402

B, T, K = 64, 1024, 6
dev = torch.device("cuda", 0)
gens = [Generator(V1, device=dev, dtype="bf16") for _ in range(2)]
params = synthetic_params(V1, 4321, "scaled")
for g in gens:
    g.load_params(params)
mel_h = torch.from_numpy(synthetic_mel(B, T, 1234)).pin_memory()
wav_h = [torch.empty((B, 256 * T), dtype=torch.float32).pin_memory() for _ in range(2)]
mel_d = [torch.empty_like(mel_h, device=dev) for _ in range(2)]
wav_d = [torch.empty((B, 256 * T), dtype=torch.float32, device=dev) for _ in range(2)]


def one(i, stream):
    with torch.cuda.stream(stream):
        mel_d[i].copy_(mel_h, non_blocking=True)
        gens[i](mel_d[i], wav_d[i])
        wav_h[i].copy_(wav_d[i], non_blocking=True)


s = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
res = {}
for name, nstream in (("resident", 0), ("serial", 1), ("two_in_flight", 2)):
    for rep in range(2):  # first repetition = warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            if nstream == 0:
                gens[0](mel_d[0], wav_d[0])
            elif nstream == 1:
                one(0, s[0])
                s[0].synchronize()
            else:
                one(k & 1, s[k & 1])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
    res[name] = {"ms_per_batch": dt * 1e3, "samples_per_s": B * 256 * T / dt}
res["bytes_per_batch"] = {"mel_h2d": mel_h.numel() * 4, "wav_d2h": B * 256 * T * 4}
print(json.dumps(res))
