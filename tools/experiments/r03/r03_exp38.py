"""long-form leg (BASELINE configs[4]): chunks per pass after the first chunk — 16 (the default so far) vs 32 / 64 / all"""
import time
import torch
from viettts_amd.hifigan.config import V1
from viettts_amd.hifigan.generator import Generator
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params
from viettts_amd.longform import synthesize_chunked

dev = torch.device("cuda:0")
g = Generator(V1, device=dev, dtype="bf16")
g.load_params(synthetic_params(V1, 4321, "scaled"))
m10 = torch.from_numpy(synthetic_mel(1, 37500, 99)[0]).to(dev)
ref = None
for rep in range(2):
    for mb in (16, 32, 64, 128):
        synthesize_chunked(g, m10, 512, max_batch=mb)  # warm-up of this shape
        torch.cuda.synchronize()
        tm = {}
        out = synthesize_chunked(g, m10, 512, max_batch=mb, timing=tm)
        if ref is None:
            ref = out.clone()
        print(f"max_batch {mb:3d}: first chunk {tm['first_chunk_s']*1e3:.2f} ms, total {tm['total_s']*1e3:.2f} ms, same bits as max_batch 16: {bool(torch.equal(out, ref))}", flush=True)
