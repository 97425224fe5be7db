"""Host-side owner of one HiFi-GAN generator on one GPU.

Mirrors the reference's ``Generator`` module (vietTTS/hifigan/model.py:77-125) as used by
``mel2wave`` (vietTTS/hifigan/mel2wave.py:28-38): construct from the hyper-parameters, hand it the
``hk_hifi.pickle`` parameter dict, call it on an NWC mel batch.  All arithmetic happens in the HIP
library (include/vtts_hifigan.h); PyTorch-ROCm only provides device memory and the stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from .. import _lib
from .config import HifiganConfig, V1
from .weights import ParamDict, check_params, conv_specs


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Generator:
    """``Generator(cfg)(mel)`` — mel ``[B, T, num_mels]`` float32 NWC -> wav ``[B, hop*T]`` float32.

    No CPU fallback: construction fails if the HIP extension is missing, and ``__call__`` fails
    unless the tensors live on a ROCm device.

    Concurrency: one Generator serves ONE call at a time.  The C handle keeps per-call state (ragged lengths, profiling
    counters, its side streams) and this object keeps one cached workspace tensor: two torch streams driving the same
    Generator concurrently would race on both.  Use one Generator per stream (the packed weights can be shared:
    ``other.adopt_packed(gen.packed_blob())``).
    """

    def __init__(self, cfg: HifiganConfig = V1, device="cuda:0", dtype: str = "f32", lib_path=None):
        self.cfg = cfg
        self.lib = _lib.load(lib_path)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("Generator needs a ROCm device ('cuda:N'); there is no CPU path")
        self.dtype = {"f32": _lib.VTTS_F32, "bf16": _lib.VTTS_BF16, "bf16x3": _lib.VTTS_BF16X3}[dtype]  # bf16x3: the fp32 engine's layouts and entry points
        self.dtype_name = dtype
        self._h = C.c_void_p(0)
        cs = _lib.make_cfg(cfg)
        dev_index = self.device.index if self.device.index is not None else 0
        _lib.check(self.lib, self.lib.vtts_hifigan_create(C.byref(cs), dev_index, self.dtype, C.byref(self._h)))
        self._blob: Optional[torch.Tensor] = None
        self._ws: Optional[torch.Tensor] = None
        self.hop = cfg.hop

    # ---- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.vtts_hifigan_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters -----------------------------------------------------------------------------
    @property
    def max_frames_per_pass(self) -> int:
        """Utterances of this many mel frames or more are refused by the C ABI (an utterance's largest activation must stay
        below 2^31 bytes: include/vtts_hifigan.h, vtts_hifigan_workspace_bytes); they go through viettts_amd.longform."""
        return self.get_option("max_frames_per_pass")  # the engine's own rule (engine.hip: check_pass_size), not a copy of it

    @property
    def packed_bytes(self) -> int:
        n = C.c_size_t(0)
        _lib.check(self.lib, self.lib.vtts_hifigan_packed_bytes(self._h, C.byref(n)))
        return int(n.value)

    def param_table(self):
        """[(key, which, shape)] the C side expects — the same 156 arrays hk_hifi.pickle holds."""
        n = C.c_int(0)
        _lib.check(self.lib, self.lib.vtts_hifigan_num_params(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            key, which = C.c_char_p(), C.c_char_p()
            shape = (C.c_int64 * 3)()
            nd = C.c_int(0)
            _lib.check(self.lib, self.lib.vtts_hifigan_param_info(self._h, i, C.byref(key), C.byref(which), shape, C.byref(nd)))
            out.append((key.value.decode(), which.value.decode(), tuple(int(shape[d]) for d in range(nd.value))))
        return out

    def _alloc_blob(self) -> torch.Tensor:
        # uint8 tensor from the caching allocator: >= 512-B aligned
        return torch.empty(self.packed_bytes, dtype=torch.uint8, device=self.device)

    def load_params(self, params: ParamDict) -> None:
        """Re-lay-out a Haiku parameter dict into the packed device blob (once, not per call as
        the reference does at mel2wave.py:35-36)."""
        check_params(self.cfg, params)
        for spec in conv_specs(self.cfg):
            for which in ("w", "b"):
                a = np.ascontiguousarray(params[spec.key][which], dtype=np.float32)
                shape = (C.c_int64 * a.ndim)(*a.shape)
                _lib.check(
                    self.lib,
                    self.lib.vtts_hifigan_set_param(self._h, spec.key.encode(), which.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim),
                )
        blob = self._alloc_blob()
        stream = torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib, self.lib.vtts_hifigan_pack(self._h, _ptr(blob), blob.numel(), C.c_void_p(stream.cuda_stream)))
        self._blob = blob

    def packed_blob(self) -> torch.Tensor:
        if self._blob is None:
            raise RuntimeError("no parameters loaded")
        return self._blob

    def adopt_packed(self, blob: torch.Tensor) -> None:
        """Bind a packed blob produced by another rank's ``load_params`` (weights broadcast once
        over RCCL; viettts_amd/dist.py)."""
        if blob.dtype != torch.uint8 or blob.numel() < self.packed_bytes or blob.device != self.device:
            raise ValueError("packed blob must be a uint8 tensor of packed_bytes on this generator's device")
        _lib.check(self.lib, self.lib.vtts_hifigan_bind_packed(self._h, _ptr(blob), blob.numel()))
        self._blob = blob

    # ---- options --------------------------------------------------------------------------------
    def set_option(self, name: str, value: int) -> None:
        _lib.check(self.lib, self.lib.vtts_hifigan_set_option(self._h, name.encode(), int(value)))
        # the cached workspace stays: _workspace() asks the engine for the size every call and only ever grows it

    def get_option(self, name: str) -> int:
        v = C.c_int64(0)
        _lib.check(self.lib, self.lib.vtts_hifigan_get_option(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    # ---- forward --------------------------------------------------------------------------------
    def workspace_bytes(self, B: int, T: int) -> int:
        n = C.c_size_t(0)
        _lib.check(self.lib, self.lib.vtts_hifigan_workspace_bytes(self._h, B, T, C.byref(n)))
        return int(n.value)

    def _workspace(self, B: int, T: int) -> torch.Tensor:
        need = self.workspace_bytes(B, T)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _check_mel(self, mel: torch.Tensor):
        if not isinstance(mel, torch.Tensor):
            raise TypeError("mel must be a torch.Tensor on the generator's device")
        if mel.device != self.device:
            raise ValueError(f"mel is on {mel.device}, generator on {self.device}")
        if mel.dtype != torch.float32 or mel.dim() != 3 or mel.shape[2] != self.cfg.num_mels:
            raise ValueError(f"mel must be float32 [B, T, {self.cfg.num_mels}] (NWC), got {tuple(mel.shape)} {mel.dtype}")
        if mel.shape[0] < 1 or mel.shape[1] < 1:
            raise ValueError("mel must have at least one utterance and one frame")
        return mel.contiguous()

    def __call__(self, mel: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Asynchronous on torch's current stream of the device."""
        mel = self._check_mel(mel)
        B, T, _ = mel.shape
        if out is None:
            out = torch.empty((B, self.hop * T), dtype=torch.float32, device=self.device)
        elif out.shape != (B, self.hop * T) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != self.device:
            raise ValueError("out must be a contiguous float32 [B, hop*T] tensor on the generator's device")
        ws = self._workspace(B, T)
        stream = torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(
                self.lib,
                self.lib.vtts_hifigan_forward(self._h, _ptr(mel), B, T, _ptr(out), _ptr(ws), ws.numel(), C.c_void_p(stream.cuda_stream)),
            )
        return out

    def forward_ragged(self, mel: torch.Tensor, frames, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Utterances of different lengths in one batch (every engine): ``mel`` ``[B, Tmax, num_mels]`` with utterance b's
        ``frames[b]`` frames at the start of its slot.  ``out[b, :hop*frames[b]]`` equals ``self(mel[b:b+1, :frames[b]])``
        bit for bit (fp32 engine: for frame counts that are multiples of 4 — an utterance run alone with another count takes the generic
        first transposed convolution, whose sums run in another order: ~1e-7); the rest of the row is zero.  ``frames``: int sequence or
        int32 tensor on the device."""
        mel = self._check_mel(mel)
        B, T, _ = mel.shape
        if isinstance(frames, torch.Tensor):
            fr = frames.to(device=self.device, dtype=torch.int32).contiguous()
            fr_host = None
        else:
            fr_host = [int(v) for v in frames]
            fr = torch.tensor(fr_host, dtype=torch.int32, device=self.device)
        if fr.numel() != B or (fr_host is not None and (min(fr_host) < 1 or max(fr_host) > T)):
            raise ValueError("frames must hold one count per utterance, 1 <= frames[b] <= mel.shape[1]")
        if out is None:
            out = torch.empty((B, self.hop * T), dtype=torch.float32, device=self.device)
        elif out.shape != (B, self.hop * T) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != self.device:
            raise ValueError("out must be a contiguous float32 [B, hop*T] tensor on the generator's device")
        ws = self._workspace(B, T)
        stream = torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(
                self.lib,
                self.lib.vtts_hifigan_forward_ragged(self._h, _ptr(mel), _ptr(fr), B, T, _ptr(out), _ptr(ws), ws.numel(), C.c_void_p(stream.cuda_stream)),
            )
        return out

    def forward_tap(self, mel: torch.Tensor, tap: str):
        """(wav, tap tensor) — test hook; tap in {"conv_pre","ups_i","mrf_i","pre_tanh"}."""
        mel = self._check_mel(mel)
        B, T, _ = mel.shape
        n = C.c_size_t(0)
        _lib.check(self.lib, self.lib.vtts_hifigan_tap_elems(self._h, tap.encode(), B, T, C.byref(n)))
        tap_t = torch.empty(int(n.value), dtype=torch.float32, device=self.device)
        out = torch.empty((B, self.hop * T), dtype=torch.float32, device=self.device)
        ws = self._workspace(B, T)
        stream = torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(
                self.lib,
                self.lib.vtts_hifigan_forward_tap(self._h, _ptr(mel), B, T, _ptr(out), _ptr(ws), ws.numel(), C.c_void_p(stream.cuda_stream),
                                                  tap.encode(), _ptr(tap_t)),
            )
        if tap == "pre_tanh":
            tap_t = tap_t.view(B, self.hop * T)
        else:
            tap_t = tap_t.view(B, -1)
        return out, tap_t

    def run_module(self, key: str, x: torch.Tensor, slope_in: float = 1.0, res: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Run one convolution module (per-layer KATs).
        fp32 handle: ``x`` is ``[B, C, L]`` channel-major (``[B, L, num_mels]`` for conv_pre) -> ``[B, Cout, Lout]``.
        bf16 handle: everything is channels-last fp32, ``x`` ``[B, L, C]`` -> ``[B, Lout, Cout]`` (``[B, Lout]`` for conv_post)."""
        spec = {s.key: s for s in conv_specs(self.cfg)}[key]
        x = x.contiguous()
        channels_last = self.dtype_name == "bf16" or key == "generator/~/conv1_d"
        if channels_last:
            B, L, _ = x.shape
        else:
            B, _, L = x.shape
        lout = L * spec.stride
        if self.dtype_name == "bf16":
            shape = (B, lout) if key == "generator/~/conv1_d_1" else (B, lout, spec.cout)
        else:
            shape = (B, spec.cout, lout)
        y = torch.empty(shape, dtype=torch.float32, device=self.device)
        if res is not None:
            res = res.contiguous()
        stream = torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(
                self.lib,
                self.lib.vtts_hifigan_run_module(self._h, key.encode(), _ptr(x), B, L, C.c_float(slope_in), _ptr(res), _ptr(y),
                                                 C.c_void_p(stream.cuda_stream)),
            )
        return y

    def run_pair(self, key_c1: str, x: torch.Tensor) -> torch.Tensor:
        """One fused ResBlock pair ``x' = convs2_z(lrelu(convs1_z(lrelu(x)))) + x`` named by its first convolution.
        bf16 handles: ``x`` fp32 ``[B, L, C]`` channels-last (rounded to bf16 on the way in) -> same shape;
        fp32 handles: ``x`` ``[B, C, L]`` channel-major -> same shape."""
        x = x.contiguous()
        if self.dtype_name == "bf16":
            B, L, _ = x.shape
        else:
            B, _, L = x.shape
        y = torch.empty_like(x)
        stream = torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib, self.lib.vtts_hifigan_run_pair(self._h, key_c1.encode(), _ptr(x), B, L, _ptr(y), C.c_void_p(stream.cuda_stream)))
        return y

    # ---- dominant-kernel timing (bench.py roofline) ----------------------------------------------
    def profile_read(self, reset: bool = True):
        ms, n, fl = C.c_double(0), C.c_int64(0), C.c_double(0)
        _lib.check(self.lib, self.lib.vtts_hifigan_profile_read(self._h, C.byref(ms), C.byref(n), C.byref(fl), int(reset)))
        name = self.lib.vtts_hifigan_profile_kernel(self._h)
        return {"ms": ms.value, "launches": int(n.value), "flops": fl.value, "kernel": name.decode() if name else ""}
