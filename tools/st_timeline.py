#!/usr/bin/env python
"""Digest the per-workgroup phase stamps of stage_bf16_k (kernels_bf16_stage.hip, development builds): mean ticks per phase over the windows that ran.
    python -m viettts_amd.csrc.build --define VTTS_TIMELINE=1 --libname libvtts_tl.so
    VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/libvtts_tl.so VTTS_ST_TL=/tmp/st.bin python tools/st_timeline.py run     (one 64 x 1024 pass of the bf16 engine, then digests)
    python tools/st_timeline.py /tmp/st.bin"""
import os
import sys

import numpy as np

PH = ["c1 interior (+taps)", "barrier", "c1 edge + fill", "epilogue 1 (0,3)", "c2 interior (+taps)", "barrier", "c2 edge + fill", "epilogue 2 (0,3) / MRF"]


def digest(path):
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, 128)
    a = a[(a[:, 0] != 0) & (a[:, 126] != 0)]
    print(f"{a.shape[0]} windows with stamps")
    t = a.astype(np.int64)
    tot = (t[:, 126] - t[:, 0]).mean()
    print(f"  {'staging (x0 -> registers -> tile A)':44s} {(t[:, 1] - t[:, 0]).mean():9.0f}")
    prev = t[:, 1]
    sums = np.zeros(8)
    for rb, k in enumerate((3, 7, 11)):
        rbsum = np.zeros(8)
        for pr in range(3):
            for i in range(8):
                cur = t[:, 2 + 24 * rb + 8 * pr + i]
                rbsum[i] += (cur - prev).mean()
                prev = cur
        print(f"  ResBlock k = {k}: " + "  ".join(f"{PH[i]} {rbsum[i] / 3:.0f}" for i in range(8)) + f"   | per pair {rbsum.sum() / 3:.0f}, MFMA floor {2 * 4 * 2 * k * 32} cycles")
        sums += rbsum
    print(f"  {'conv_post + tanh':44s} {(t[:, 126] - prev).mean():9.0f}")
    print(f"  whole window {tot:.0f} ticks; MFMA floor {2 * 4 * 2 * 21 * 3 * 32} cycles per wave")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import torch

        from viettts_amd.hifigan.config import V1
        from viettts_amd.hifigan.generator import Generator
        from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params

        g = Generator(V1, device="cuda:0", dtype="bf16")
        g.load_params(synthetic_params(V1, 4321, "scaled"))
        g.set_option("streams", 1)
        g.set_option("microbatch", 64)
        mel = torch.from_numpy(synthetic_mel(64, 1024, 1234)).to("cuda:0")
        g(mel)
        g(mel)
        torch.cuda.synchronize()
        digest(os.environ["VTTS_ST_TL"])
    else:
        digest(sys.argv[1])
