"""Bisect: which step of bench.py's prologue makes the pipeline leg slow?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench
from viettts_amd import dist as vdist
from viettts_amd.hifigan.config import V1
from viettts_amd.hifigan.generator import Generator
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params

def show(tag, r):
    print(tag, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if k in ("generator_ms", "total_ms", "pinned_alloc_ms")}, flush=True)

mode = sys.argv[1]
if mode != "nosetdev":
    torch.cuda.set_device(0)
dev = torch.device("cuda", 0) if mode != "nosetdev" else "cuda:0"
gen = Generator(V1, device=dev, dtype="bf16")

if mode == "dp":
    vdist.setup_generator_dp(gen, lambda: synthetic_params(V1, 4321, "scaled"), vdist.rank_info(), {})
else:
    gen.load_params(synthetic_params(V1, 4321, "scaled"))
if mode == "pipefirst":
    show("pipefirst: before the big passes", bench.pipeline_256(256, gen))
mel = torch.from_numpy(synthetic_mel(64, 1024, 1234)).to(dev)
out = torch.empty((64, 256 * 1024), dtype=torch.float32, device=dev)
for _ in range(0 if mode == "nobig" else 3):
    gen(mel, out)
torch.cuda.synchronize()
if mode == "smallfirst":  # a small generator call before the pipeline
    gen(mel[:2, :200].contiguous()); torch.cuda.synchronize()
if mode == "opts":
    gen.set_option("streams", 1); gen.set_option("microbatch", 64); gen(mel, out); torch.cuda.synchronize(); gen.set_option("streams", 0); gen.set_option("microbatch", 0)
if mode == "profile":
    gen.set_option("profile", 1); gen.profile_read(reset=True); gen(mel, out); torch.cuda.synchronize(); gen.profile_read(reset=True); gen.set_option("profile", 0)
if mode == "barrier":
    show(mode, bench.pipeline_256(256, gen, 0, 1, lambda: None)); sys.exit(0)
show(mode, bench.pipeline_256(256, gen))
