#!/usr/bin/env python
"""Digest the per-workgroup phase stamps of resblock_x3_k (kernels_x3_rb.hip, VTTS_RX_TL=<file>): mean ticks per phase over the workgroups that ran.
    python -m viettts_amd.csrc.build --define VTTS_TIMELINE=1 --libname libvtts_tl.so      (the stamps are compiled into development builds only)
    VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/libvtts_tl.so VTTS_RX_TL=/tmp/tl.bin VTTS_RX_TL_K=3 python tools/rx_timeline.py run      (runs one 64 x 1024 pass on the split engine, then digests)
    python tools/rx_timeline.py /tmp/tl.bin"""
import os
import sys

import numpy as np

NAMES = ["load x + stage", "barrier"]
for pr in range(3):
    NAMES += [f"p{pr} c1 MFMA", f"p{pr} barrier", f"p{pr} epilogue 1 (+barrier)", f"p{pr} c2 MFMA"]
    if pr < 2:
        NAMES += [f"p{pr} barrier", f"p{pr} residual + A write (+barrier)"]
STAMPS = [0, 1, 2] + [3 + 6 * pr + i for pr in range(3) for i in range(6) if not (pr == 2 and i >= 4)]


def digest(path):
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, 24)
    a = a[(a[:, 0] != 0) & (a[:, 22] != 0)]
    print(f"{a.shape[0]} workgroups with stamps")
    t = a[:, STAMPS + [22]].astype(np.int64)
    d = np.diff(t, axis=1)
    names = NAMES + ["MRF bookkeeping + store"]
    tot = (t[:, -1] - t[:, 0]).mean()
    for n, v in zip(names, d.mean(axis=0)):
        print(f"  {n:40s} {v:9.0f} ticks  {100 * v / tot:5.1f} %")
    print(f"  {'whole window':40s} {tot:9.0f} ticks (s_memtime counts shader clocks on gfx950: ~1.9 GHz under this load -> {tot / 1900:.1f} us)")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import torch

        from viettts_amd.hifigan.config import V1
        from viettts_amd.hifigan.generator import Generator
        from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params

        g = Generator(V1, device="cuda:0", dtype="bf16x3")
        g.load_params(synthetic_params(V1, 4321, "scaled"))
        g.set_option("streams", 1)
        g.set_option("microbatch", 64)
        g.set_option("fuse", 3)
        mel = torch.from_numpy(synthetic_mel(64, 1024, 1234)).to("cuda:0")
        g(mel)
        g(mel)
        torch.cuda.synchronize()
        digest(os.environ["VTTS_RX_TL"])
    else:
        digest(sys.argv[1])
