"""Round 4, VERDICT item 5 — numerical half of the go / no-go: can a parity-grade (<= 1e-4 max-abs vs the reference) generator run on the bf16
matrix pipe with SPLIT operands?  CPU emulation (test infrastructure: it drives oracle/hifigan_oracle.py with its two convolution primitives
replaced), no GPU involved.

Every convolution's operands (activations as the engine would hold them — fp32 in HBM — and weights) are split into bf16 terms
    v = v0 + v1 (+ v2),   v0 = bf16(v), v1 = bf16(v - v0), v2 = bf16(v - v0 - v1)
and the product is formed from bf16 x bf16 terms only, accumulated in fp32 (what v_mfma_f32_32x32x16_bf16 does):
    bf16     : x0 w0                                              1 MFMA per product   (the bf16 engine's arithmetic, fp32 storage)
    split3   : x0 w0 + x0 w1 + x1 w0                              3 MFMAs   (~16 significand bits per operand)
    split4   : split3 + x1 w1                                     4
    split6   : x0w0 + x0w1 + x1w0 + x0w2 + x1w1 + x2w0            6         (~24 bits: fp32-grade)
Everything else (bias, LeakyReLU, residual, MRF sum / mean, tanh) in fp32.  Reported: max-abs of the waveform and of the pre-tanh signal against the
fp64 oracle on the benchmark's own synthetic weights / mels (W_scaled 4321, mel seed 1234), next to a plain fp32 run of the same code (the
fp32 engine's error level).

    python tools/experiments/r04/split_precision_emulation.py [T=64] [B=2]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle import hifigan_oracle as orc  # noqa: E402
from viettts_amd.hifigan.config import V1  # noqa: E402
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params  # noqa: E402


def bf16(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split(v, n):
    v = np.asarray(v, dtype=np.float32)
    out, r = [], v.copy()
    for _ in range(n):
        t = bf16(r)
        out.append(t)
        r = (r - t).astype(np.float32)
    return out


TERMS = {
    "bf16": (1, [(0, 0)]),
    "split3": (2, [(0, 0), (0, 1), (1, 0)]),
    "split4": (2, [(0, 0), (0, 1), (1, 0), (1, 1)]),
    "split6": (3, [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]),
}


def make_prims(mode):
    nsplit, terms = TERMS[mode]
    wcache = {}

    def wsplit(w):
        key = id(w)
        if key not in wcache:
            wcache[key] = (w, split(w, nsplit))
        return wcache[key][1]

    def conv1d(x, w, b, rate, pad):
        B, T, cin = x.shape
        k, _, cout = w.shape
        xs = split(x, nsplit)
        ws = wsplit(w)
        t_out = T + 2 * pad - (k - 1) * rate
        y = np.empty((B, t_out, cout), dtype=np.float32)
        y[...] = b.astype(np.float32)
        xp = []
        for xi in xs:
            p = np.zeros((B, T + 2 * pad, cin), dtype=np.float32)
            p[:, pad : pad + T] = xi
            xp.append(p)
        for j in range(k):
            for (ia, iw) in terms[::-1]:  # small terms first, as a kernel would order them
                y += xp[ia][:, j * rate : j * rate + t_out] @ ws[iw][j]
        return y

    def conv1d_transpose(x, w, b, stride):
        B, T, cin = x.shape
        k, cout, _ = w.shape
        pa, pb = orc.conv_transpose_same_pads(k, stride)
        ld = (T - 1) * stride + 1
        xs = split(x, nsplit)
        ws = wsplit(w)
        t_out = ld + pa + pb - k + 1
        y = np.empty((B, t_out, cout), dtype=np.float32)
        y[...] = b.astype(np.float32)
        xd = []
        for xi in xs:
            d = np.zeros((B, ld + pa + pb, cin), dtype=np.float32)
            d[:, pa : pa + ld : stride] = xi
            xd.append(d)
        for j in range(k):
            for (ia, iw) in terms[::-1]:
                y += xd[ia][:, j : j + t_out] @ ws[iw][j].T
        return y

    return conv1d, conv1d_transpose


def run(mode, params, mel):
    c1, ct = orc.conv1d, orc.conv1d_transpose
    try:
        if mode != "fp32":
            orc.conv1d, orc.conv1d_transpose = make_prims(mode)
        y, pre = orc.generator_forward(params, mel, V1, np.float32, return_pre_tanh=True)
    finally:
        orc.conv1d, orc.conv1d_transpose = c1, ct
    return y[..., 0].astype(np.float64), pre[..., 0].astype(np.float64)


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    params = synthetic_params(V1, 4321, "scaled")
    mel = synthetic_mel(B, T, 1234)
    y64, p64 = orc.generator_forward(params, mel, V1, np.float64, return_pre_tanh=True)
    y64, p64 = y64[..., 0], p64[..., 0]
    print(f"HiFi-GAN V1, W_scaled(4321), mel seed 1234, B={B} x T={T} ({y64.size} samples); |pre-tanh| max {np.abs(p64).max():.2f}")
    print(f"{'mode':8s} {'MFMAs/product':>13s} {'max|dy|':>11s} {'max|dpre|':>11s} {'rms dpre':>11s} {'SNR pre dB':>10s}")
    for mode in ("fp32", "bf16", "split3", "split4", "split6"):
        y, p = run(mode, params, mel)
        n = {"fp32": "(fp32 MFMA)"}.get(mode) or str(len(TERMS[mode][1]))
        snr = 10 * np.log10((p64 ** 2).mean() / ((p - p64) ** 2).mean())
        print(f"{mode:8s} {n:>13s} {np.abs(y - y64).max():11.3e} {np.abs(p - p64).max():11.3e} {np.sqrt(((p - p64) ** 2).mean()):11.3e} {snr:10.1f}")


if __name__ == "__main__":
    main()
