"""long-form leg with option "zigzag" 0 / 1 (a regression check)"""
import torch
from viettts_amd.hifigan.config import V1
from viettts_amd.hifigan.generator import Generator
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params
from viettts_amd.longform import synthesize_chunked

dev = torch.device("cuda:0")
g = Generator(V1, device=dev, dtype="bf16")
g.load_params(synthetic_params(V1, 4321, "scaled"))
m10 = torch.from_numpy(synthetic_mel(1, 37500, 99)[0]).to(dev)
for rep in range(3):
    for z in (0, 1):
        g.set_option("zigzag", z)
        synthesize_chunked(g, m10, 512)
        torch.cuda.synchronize()
        tm = {}
        synthesize_chunked(g, m10, 512, timing=tm)
        print(f"zigzag {z}: first chunk {tm['first_chunk_s']*1e3:.2f} ms, total {tm['total_s']*1e3:.2f} ms", flush=True)
