"""Drop-in for ``vietTTS.hifigan.mel2wave.mel2wave`` (vietTTS/hifigan/mel2wave.py:20-41).

Same name, same positional signature, same files read (``assets/hifigan/config.json`` relative to
the CWD, ``./assets/infore/hifigan/hk_hifi.pickle``), same result (host ``np.float32`` array,
``jnp.squeeze``-d: ``[256*T]`` for a single utterance, ``[B, 256*T]`` for a batch).  Differences,
none of them observable in the output: the config and the 55.7 MB pickle are read once and cached
(the reference re-reads both on every call; ``reload()`` drops the cache), and the generator runs on
MI355X through the HIP library instead of un-jitted XLA-CPU ops.

Engine: the drop-in runs the **fp32** engine by default — the one that meets the reference's numerics (<= 1e-4 max-abs
against the Haiku generator, BASELINE.json; measured 1.5e-6) at ~5.0e7 samples/s.  ``VTTS_MEL2WAVE_DTYPE=bf16`` in the environment (or
``FLAGS.dtype = "bf16"``) selects the bf16 throughput engine (~4e8 samples/s batched, max-abs ~1e-2 / 45 dB SNR against
the same reference: bench.py ``parity_bf16``); ``bf16x3`` the split-operand engine (the fp32 engine's layouts with the ResBlock
convolutions on the bf16 matrix pipe, three bf16 products per term: 1.5e-5 against the same reference at 64 x 1024 frames — inside the 1e-4 bar — at
~3.0x the fp32 engine's batched throughput: bench.py ``bf16x3_path.parity`` / ``.speedup_vs_fp32_engine``, BENCH_r05.json).

Errors follow the reference: a missing config / checkpoint raises ``FileNotFoundError``; a wrong
mel shape raises ``ValueError``.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from .config import FLAGS, HifiganConfig
from .generator import Generator
from .weights import load_haiku_pickle

_CACHE: dict = {}


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("mel2wave needs an MI355X visible to PyTorch-ROCm; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def reload() -> None:
    """Forget cached config/weights (the reference has no cache: it re-reads per call)."""
    for g in _CACHE.values():
        g.close()
    _CACHE.clear()


def _engine_dtype() -> str:
    d = os.environ.get("VTTS_MEL2WAVE_DTYPE") or getattr(FLAGS, "dtype", None) or "f32"
    if d not in ("f32", "bf16", "bf16x3"):
        raise ValueError(f"VTTS_MEL2WAVE_DTYPE / FLAGS.dtype must be 'f32', 'bf16' or 'bf16x3', got {d!r}")
    return d


def _generator(dtype: Optional[str] = None) -> Generator:
    dtype = dtype or _engine_dtype()
    config_file = str(FLAGS.config_file)  # CWD-relative, as mel2wave.py:21
    ckpt = FLAGS.ckpt_dir / "hk_hifi.pickle"  # mel2wave.py:35
    if not os.path.exists(config_file):
        raise FileNotFoundError(config_file)
    if not os.path.exists(ckpt):
        raise FileNotFoundError(str(ckpt))
    dev = _device()
    key = (os.path.abspath(config_file), os.path.getmtime(config_file), os.path.abspath(ckpt), os.path.getmtime(ckpt), str(dev), dtype)
    g = _CACHE.get(key)
    if g is None:
        cfg = HifiganConfig.from_json(config_file)
        g = Generator(cfg, device=dev, dtype=dtype)
        g.load_params(load_haiku_pickle(ckpt))
        _CACHE[key] = g
    return g


def mel2wave(mel):
    """mel: ``[B, T, 80]`` float32 log-mel, NWC (numpy array, torch tensor, or anything
    ``np.asarray`` accepts) -> waveform, float32 numpy, squeezed."""
    g = _generator()
    if isinstance(mel, torch.Tensor):
        m = mel.detach().to(device=g.device, dtype=torch.float32)
    else:
        m = torch.from_numpy(np.ascontiguousarray(np.asarray(mel), dtype=np.float32)).to(g.device)
    if m.dim() != 3:
        raise ValueError(f"mel must be [B, T, {g.cfg.num_mels}], got shape {tuple(m.shape)}")
    if m.shape[1] >= g.max_frames_per_pass:
        # longer than one pass of the kernels takes (the reference has no such limit: one shot, memory permitting): exact
        # chunking with the receptive field as halo — the interior equals the one-shot result up to fp32 reassociation
        from ..longform import synthesize_chunked

        wav = torch.stack([synthesize_chunked(g, m[b].contiguous(), chunk_frames=4096, max_batch=8) for b in range(m.shape[0])])
    else:
        wav = g(m)
    # jnp.squeeze + jax.device_get (mel2wave.py:39-40); .cpu() synchronises the stream
    return np.squeeze(wav.cpu().numpy())
