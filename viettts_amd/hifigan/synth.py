"""Deterministic synthetic weights and mel inputs.

No pretrained checkpoint exists offline (the reference fetches them with wget,
scripts/quick_start.sh:4-6), so parity and throughput runs use seeded random weights
of the real architecture.  Everything here is drawn from a *CPU* ``torch.Generator`` so
that the same seeds give the same bits on every box with this image; the golden
fixtures under tests/golden/ additionally pin a SHA-256 of the resulting blobs.

``W_scaled`` (primary): modules visited in upstream ``named_modules()`` order; for each
module first the weight (in upstream/torch layout), then the bias:
  Conv1d          w ~ N(0, 1/(Cin*k))
  ConvTranspose1d w ~ N(0, stride/(Cin*k))
  bias            b ~ N(0, 0.01**2)
  conv_post weight additionally * 0.2 (keeps the pre-tanh signal out of saturation).
``W_init``: the reference's own initialiser N(0, 0.01) on every weight
(vietTTS/hifigan/torch_model.py:16-19), same visiting order, bias as above.

Mel ``M(B,T,seed)``: ``clamp(-5 + 2*randn(B,T,80), log(1e-5), 2.0)``, NWC float32 — the
range of real log-mels (vietTTS/nat/dsp.py:127).
"""
from __future__ import annotations

import hashlib
import math

import numpy as np
import torch

from .config import HifiganConfig, V1
from .weights import ParamDict, specs_named_modules_order, torch_weight_to_haiku

MEL_FLOOR = math.log(1e-5)  # -11.512925...


def synthetic_params(cfg: HifiganConfig = V1, seed: int = 4321, kind: str = "scaled") -> ParamDict:
    """Haiku-layout parameter dict of synthetic weights (see module docstring)."""
    if kind not in ("scaled", "init"):
        raise ValueError("kind must be 'scaled' or 'init'")
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    out: ParamDict = {}
    for spec in specs_named_modules_order(cfg):
        shape = spec.torch_w_shape
        w = torch.randn(shape, generator=g, dtype=torch.float32)
        if kind == "scaled":
            fan = spec.cin * spec.k
            var = (spec.stride / fan) if spec.kind == "convT" else (1.0 / fan)
            w = w * math.sqrt(var)
            if spec.torch_prefix == "conv_post":
                w = w * 0.2
        else:
            w = w * 0.01
        b = torch.randn((spec.cout,), generator=g, dtype=torch.float32) * 0.01
        out[spec.key] = {"w": torch_weight_to_haiku(spec, w.numpy()), "b": b.numpy().copy()}
    return out


def synthetic_mel(B: int, T: int, seed: int = 1234, num_mels: int = 80) -> np.ndarray:
    """``[B, T, num_mels]`` float32 NWC log-mel-like input."""
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    m = -5.0 + 2.0 * torch.randn((B, T, num_mels), generator=g, dtype=torch.float32)
    return torch.clamp(m, MEL_FLOOR, 2.0).numpy().copy()


def params_digest(params: ParamDict) -> str:
    """SHA-256 over the sorted-key concatenation of every array's bytes."""
    h = hashlib.sha256()
    for k in sorted(params):
        for n in ("w", "b"):
            a = np.ascontiguousarray(params[k][n], dtype=np.float32)
            h.update(k.encode())
            h.update(n.encode())
            h.update(str(a.shape).encode())
            h.update(a.tobytes())
    return h.hexdigest()


def array_digest(a: np.ndarray) -> str:
    a = np.ascontiguousarray(a)
    return hashlib.sha256(str(a.dtype).encode() + str(a.shape).encode() + a.tobytes()).hexdigest()
