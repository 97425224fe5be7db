"""Why is pipeline_256 slower inside bench.py (86 ms) than standalone (64 ms)?  Same process: fresh generator vs the generator that just ran 64 x 1024 passes."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench
from viettts_amd.hifigan.config import V1
from viettts_amd.hifigan.generator import Generator
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params

def show(tag, r):
    print(tag, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if k.endswith("_ms") or k == "overlap_groups"}, flush=True)

show("fresh generator, groups default", bench.pipeline_256(256))
show("fresh generator, groups 1", bench.pipeline_256(256, overlap_groups=1))
gen = Generator(V1, device="cuda:0", dtype="bf16")
gen.load_params(synthetic_params(V1, 4321, "scaled"))
show("own generator before any big pass", bench.pipeline_256(256, gen))
mel = torch.from_numpy(synthetic_mel(64, 1024, 1234)).to("cuda:0")
out = torch.empty((64, 256 * 1024), dtype=torch.float32, device="cuda:0")
for _ in range(3):
    gen(mel, out)
torch.cuda.synchronize()
show("same generator after 64 x 1024 passes", bench.pipeline_256(256, gen))
show("same generator after 64 x 1024 passes, groups 1", bench.pipeline_256(256, gen, overlap_groups=1))
gen._ws = None
torch.cuda.empty_cache()
show("same generator, workspace dropped", bench.pipeline_256(256, gen))
del mel, out
torch.cuda.empty_cache()
show("... and the big tensors freed", bench.pipeline_256(256, gen))
