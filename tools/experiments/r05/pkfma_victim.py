"""Round 5: a SELF-CHECKING packed-f32 victim (tools/kbench/pkfma_victim.hip: nat.hip's `partial` loop as v_pk_fma_f32 and as scalar v_fma_f32 in the
same thread, compared bit for bit) beside the real generators.  Usage: python pkfma_victim.py [reps]"""
import ctypes as C
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R)
from viettts_amd import _lib
from viettts_amd.hifigan.config import V1
from viettts_amd.hifigan.generator import Generator
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params

_lib.load()  # binds the HIP runtime torch uses
vic = C.CDLL(os.path.join(R, "tools", "kbench", "bin", "libpkfma_victim.so"))
vic.pkfma_victim_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
params = synthetic_params(V1, 4321, "scaled")
gens = {}
for dt in ("bf16", "f32", "bf16x3"):
    g = Generator(V1, device=dev, dtype=dt)
    g.load_params(params)
    gens[dt] = g
other = torch.from_numpy(synthetic_mel(8, 200, 5)).to(dev)
big = torch.from_numpy(synthetic_mel(64, 256, 6)).to(dev)
for g in gens.values():
    g(other); g(big)
torch.cuda.synchronize()
rows = 1024
weights = torch.randn(rows // 4 * 1024 * 4, device=dev) * 0.5  # [rows / 4][THREADS][4]
counts = torch.zeros(4, dtype=torch.int32, device=dev)
cur = torch.cuda.current_stream()
s_v = torch.cuda.Stream(priority=-1)


def trial(name, variant, beside, mel, wgs=3, iters=40, launches=300):
    tot = [0, 0, 0]
    for rep in range(reps):
        counts.zero_()
        torch.cuda.synchronize()
        s_v.wait_stream(cur)
        with torch.cuda.stream(s_v):
            for _ in range(launches):
                rc = vic.pkfma_victim_launch(C.c_void_p(s_v.cuda_stream), variant, wgs, iters, rows, C.c_void_p(weights.data_ptr()), C.c_void_p(counts.data_ptr()))
                assert rc == 0, rc
        if beside is not None:
            torch.cuda._sleep(int(2.0e6))
            for _ in range(6):
                gens[beside](mel)
        cur.wait_stream(s_v)
        torch.cuda.synchronize()
        c = counts.cpu().tolist()
        for i in range(3):
            tot[i] += c[i]
    print(f"{name:58s} beside {str(beside):7s}: LOW-half mismatches {tot[0]:9d}, HIGH-half {tot[1]:6d}  ({tot[2]} workgroups x {iters} passes of {rows} rows)", flush=True)


V = {"w global, 1024 thr, x LDS (as nat.hip)": 3, "w LDS, 1024 thr, x LDS": 2, "w global, 256 thr, x LDS": 1, "w LDS, 256 thr, x LDS": 0, "w global, 1024 thr, x regs": 7, "w LDS, 256 thr, x regs": 4}
for name, v in V.items():
    for beside in ("bf16", None, "f32", "bf16x3"):
        trial(name, v, beside, other)
trial("w global, 1024 thr, x LDS; 64 workgroups", 3, "bf16", big, wgs=64, iters=10, launches=100)
trial("w global, 1024 thr, x LDS; 64 workgroups", 3, None, big, wgs=64, iters=10, launches=100)
