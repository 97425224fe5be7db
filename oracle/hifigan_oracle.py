"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU (numpy) restatement of the reference's mel->waveform algorithm: the Haiku HiFi-GAN
generator of NTT123/vietTTS, taking the *Haiku-layout* parameter dict (``hk_hifi.pickle``).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product (viettts_amd/) never does.

Pinning status: the reference's own tests hold NO golden vector for this path
 and its JAX/Haiku implementation cannot run offline (jax, jaxlib,
dm-haiku absent).  The oracle is therefore pinned against outputs of the reference's own
PyTorch generator (vietTTS/hifigan/torch_model.py:156-218 — the model the Haiku weights
are converted from) run in the build container by oracle/make_golden.py, through the
reference's own converter (vietTTS/hifigan/convert_torch_model_to_haiku.py:27-62); those
outputs are committed under tests/golden/.  Since round 2 the same fixtures also hold ``y64_haiku``: the output of the
reference's HAIKU generator itself — vietTTS/hifigan/model.py + mel2wave.py imported from /root/reference and executed
unchanged over oracle/haiku_shim.py (whose convolutions are the two primitives below) — which equals the torch generator's to
2e-15: the wiring and module names of the module the product replaces are the reference's, executed; the primitives are pinned
numerically by the torch twin through the converter's layout map.  Third-party arithmetic restated here:
dm-haiku ``hk.Conv1D`` / ``hk.Conv1DTranspose`` and jax ``lax.conv_general_dilated`` /
``lax.conv_transpose`` / ``jax.nn.leaky_relu`` / ``jnp.tanh`` (unpinned in the reference's
setup.py:6-19).

Layout is the reference's: activations NWC ``[B, time, channels]``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np

LRELU_SLOPE = 0.1  # vietTTS/hifigan/model.py:5
FINAL_LRELU_SLOPE = 0.01  # jax.nn.leaky_relu default, vietTTS/hifigan/model.py:122


def get_padding(kernel_size: int, dilation: int = 1) -> int:
    """vietTTS/hifigan/model.py:8-10 — symmetric explicit pad ``int((k*d - d)/2)``."""
    return int((kernel_size * dilation - dilation) / 2)


def leaky_relu(x: np.ndarray, slope: float) -> np.ndarray:
    """jax.nn.leaky_relu: ``where(x >= 0, x, slope*x)``.  The slope is rounded to x.dtype
    first, as XLA/torch do for an fp32 tensor."""
    s = x.dtype.type(slope)
    return np.where(x >= 0, x, s * x)


def conv1d(x: np.ndarray, w: np.ndarray, b: np.ndarray, rate: int, pad: int) -> np.ndarray:
    """hk.Conv1D(stride=1, rate=d, padding=((p,p),)) — SURVEY.md Appendix A.1.

    ``y[b,t,co] = bias[co] + sum_j sum_ci w[j,ci,co] * xpad[b, t + j*d, ci]``
    (cross-correlation, no kernel flip).  x ``[B,T,Cin]``, w ``[K,Cin,Cout]``.
    Call sites: model.py:21-28 (convs1), :33-40 (convs2), :83 (conv_pre), :107 (conv_post).
    """
    B, T, cin = x.shape
    k, cin_w, cout = w.shape
    assert cin == cin_w, (cin, cin_w)
    xp = np.zeros((B, T + 2 * pad, cin), dtype=x.dtype)
    xp[:, pad : pad + T] = x
    t_out = T + 2 * pad - (k - 1) * rate
    y = np.empty((B, t_out, cout), dtype=x.dtype)
    y[...] = b.astype(x.dtype)
    for j in range(k):
        y += xp[:, j * rate : j * rate + t_out] @ w[j].astype(x.dtype)
    return y


def conv_transpose_same_pads(k: int, s: int) -> Tuple[int, int]:
    """lax.conv_transpose padding="SAME" (jax _conv_transpose_padding): with
    ``pad_len = k + s - 2``: ``pad_a = k-1 if s > k-1 else ceil(pad_len/2)``.
    (16,8) -> (11,11); (4,2) -> (2,2).  SURVEY.md Appendix A.2."""
    pad_len = k + s - 2
    pad_a = (k - 1) if s > k - 1 else int(math.ceil(pad_len / 2))
    return pad_a, pad_len - pad_a


def conv1d_transpose(x: np.ndarray, w: np.ndarray, b: np.ndarray, stride: int) -> np.ndarray:
    """hk.Conv1DTranspose(stride=s, padding="SAME") — model.py:88-94, SURVEY.md A.2.

    Zero-stuff the input (``xd[s*t] = x[t]``), pad (pad_a, pad_b), then correlate *without*
    flipping: ``y[b,p,co] = bias[co] + sum_j sum_ci w[j,co,ci] * xdpad[b, p+j, ci]``.
    w ``[K,Cout,Cin]``; output length ``s*T``.
    """
    B, T, cin = x.shape
    k, cout, cin_w = w.shape
    assert cin == cin_w
    pa, pb = conv_transpose_same_pads(k, stride)
    ld = (T - 1) * stride + 1
    xd = np.zeros((B, ld + pa + pb, cin), dtype=x.dtype)
    xd[:, pa : pa + ld : stride] = x
    t_out = ld + pa + pb - k + 1
    assert t_out == stride * T, (t_out, stride * T)
    y = np.empty((B, t_out, cout), dtype=x.dtype)
    y[...] = b.astype(x.dtype)
    for j in range(k):
        y += xd[:, j : j + t_out] @ w[j].astype(x.dtype).T
    return y


def resblock1(params: Dict, n: int, x: np.ndarray, k: int, dilations) -> np.ndarray:
    """ResBlock1.__call__ — vietTTS/hifigan/model.py:44-51."""
    for z, d in enumerate(dilations):
        c1 = params[f"generator/~/res_block1_{n}/~/convs1_{z}"]
        c2 = params[f"generator/~/res_block1_{n}/~/convs2_{z}"]
        xt = leaky_relu(x, LRELU_SLOPE)
        xt = conv1d(xt, c1["w"], c1["b"], d, get_padding(k, d))
        xt = leaky_relu(xt, LRELU_SLOPE)
        xt = conv1d(xt, c2["w"], c2["b"], 1, get_padding(k, 1))
        x = xt + x
    return x


def resblock2(params: Dict, n: int, x: np.ndarray, k: int, dilations) -> np.ndarray:
    """ResBlock2.__call__ — vietTTS/hifigan/model.py:69-74: per convolution ``x = c(leaky_relu(x)) + x``.  Module names as the
    Haiku model creates them: ``res_block1_{n}`` (model.py:105) with default-named convolutions ``conv1_d``, ``conv1_d_1``."""
    for z, d in enumerate(dilations):
        c = params[f"generator/~/res_block1_{n}/~/conv1_d" + ("" if z == 0 else f"_{z}")]
        xt = leaky_relu(x, LRELU_SLOPE)
        xt = conv1d(xt, c["w"], c["b"], int(d), get_padding(k, int(d)))
        x = xt + x
    return x


def generator_forward(params: Dict, mel: np.ndarray, cfg=None, dtype=np.float64, return_pre_tanh: bool = False, taps: Optional[List] = None):
    """Generator.__call__ — vietTTS/hifigan/model.py:109-125.

    ``mel`` is ``[B, T, num_mels]`` NWC; returns ``[B, hop*T, 1]`` (and the pre-tanh tensor
    if asked).  ``taps``, if a list, receives ``(name, array)`` for every intermediate the
    per-layer parity tests compare.
    """
    if cfg is None:
        from viettts_amd.hifigan.config import V1 as cfg  # noqa: N811
    p = {k: {n: np.asarray(a, dtype=dtype) for n, a in m.items()} for k, m in params.items()}
    x = np.asarray(mel, dtype=dtype)
    pre = p["generator/~/conv1_d"]
    x = conv1d(x, pre["w"], pre["b"], 1, 3)  # model.py:110
    if taps is not None:
        taps.append(("conv_pre", x))
    nk = len(cfg.resblock_kernel_sizes)
    for i in range(len(cfg.upsample_rates)):
        x = leaky_relu(x, LRELU_SLOPE)  # :112
        up = p[f"generator/~/ups_{i}"]
        x = conv1d_transpose(x, up["w"], up["b"], int(cfg.upsample_rates[i]))  # :114
        if taps is not None:
            taps.append((f"ups_{i}", x))
        xs = None
        for j in range(nk):  # :116-120
            rb = resblock2 if getattr(cfg, "resblock", "1") == "2" else resblock1  # model.py:86
            r = rb(p, i * nk + j, x, int(cfg.resblock_kernel_sizes[j]), cfg.resblock_dilation_sizes[j])
            if taps is not None:
                taps.append((f"res_block1_{i * nk + j}", r))
            xs = r if xs is None else xs + r
        x = xs / dtype(nk)  # :121 true division
        if taps is not None:
            taps.append((f"mrf_{i}", x))
    x = leaky_relu(x, FINAL_LRELU_SLOPE)  # :122
    post = p["generator/~/conv1_d_1"]
    x = conv1d(x, post["w"], post["b"], 1, 3)  # :123
    y = np.tanh(x)  # :124
    if return_pre_tanh:
        return y, x
    return y


def mel2wave_oracle(params: Dict, mel: np.ndarray, cfg=None, dtype=np.float64) -> np.ndarray:
    """The boundary contract of vietTTS/hifigan/mel2wave.py:37-41: squeeze -> host float32."""
    y = generator_forward(params, mel, cfg, dtype)
    return np.squeeze(y).astype(np.float32)


def flops_per_frame(cfg) -> int:
    """2*MAC of all convolutions per mel frame: 614 105 088 for V1."""
    from viettts_amd.hifigan.weights import conv_specs

    total = 0
    length = 1  # positions per mel frame at the conv's *output*
    for s in conv_specs(cfg):
        if s.kind == "convT":
            # every input position meets all k taps of every (ci, co): MAC = L_in*cin*cout*k
            total += length * s.cin * s.cout * s.k
            length *= s.stride
        else:
            total += length * s.cin * s.cout * s.k
    return 2 * total
