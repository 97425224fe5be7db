"""Host-side owner of the NAT duration model on one GPU.

Mirrors ``predict_duration(tokens)`` of the reference (vietTTS/nat/text2mel.py:22-34): build
``DurationModel(is_training=False)`` (vietTTS/nat/model.py:53-70), hand it the checkpoint's ``params`` and ``aux``
dicts, apply it to token ids.  All arithmetic happens in the HIP library (include/vtts_nat.h); PyTorch-ROCm only
provides device memory and the stream.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import _lib
from .config import FLAGS

HaikuDict = Dict[str, Dict[str, np.ndarray]]


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _lookup(d: HaikuDict, tail: str, name: str) -> np.ndarray:
    """An array by the tail of its Haiku module path (the leading scope depends on how the reference wrapped the
    module in hk.transform: ``duration_model/~/...`` for text2mel.py:23-24)."""
    hits = [k for k in d if (k == tail or k.endswith("/" + tail)) and name in d[k]]
    # "linear" also ends "lstm/linear": keep the shortest path, i.e. the module directly under the model scope
    hits.sort(key=len)
    hits = [k for k in hits if len(k) == len(hits[0])] if hits else hits
    if len(hits) != 1:
        raise KeyError(f"checkpoint has {len(hits)} modules ending in {tail!r} with an array {name!r}")
    return np.ascontiguousarray(d[hits[0]][name], dtype=np.float32)


class DurationModel:
    """``DurationModel()(tokens_list) -> [seconds per token]`` for a batch of sentences (the reference runs them one
    at a time; rows are independent)."""

    def __init__(self, vocab_size: int = FLAGS.vocab_size, lstm_dim: int = FLAGS.duration_lstm_dim, device="cuda:0", lib_path=None):
        self.lib = _lib.load(lib_path)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("DurationModel needs a ROCm device ('cuda:N'); there is no CPU path")
        self.vocab_size, self.lstm_dim = int(vocab_size), int(lstm_dim)
        self._h = C.c_void_p(0)
        cfg = _lib.NatDurationCfg(self.vocab_size, self.lstm_dim)
        dev_index = self.device.index if self.device.index is not None else 0
        _lib.check(self.lib, self.lib.vtts_nat_duration_create(C.byref(cfg), dev_index, C.byref(self._h)))
        self._blob: Optional[torch.Tensor] = None
        self._ws: Optional[torch.Tensor] = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.vtts_nat_duration_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def param_table(self):
        """[(module tail, array name, shape)] the C side expects."""
        n = C.c_int(0)
        _lib.check(self.lib, self.lib.vtts_nat_duration_num_params(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            mod, name = C.c_char_p(), C.c_char_p()
            shape = (C.c_int64 * 3)()
            nd = C.c_int(0)
            _lib.check(self.lib, self.lib.vtts_nat_duration_param_info(self._h, i, C.byref(mod), C.byref(name), shape, C.byref(nd)))
            out.append((mod.value.decode(), name.value.decode(), tuple(int(shape[d]) for d in range(nd.value))))
        return out

    # ---- packed weights: rank 0 packs, the other ranks of a data-parallel job receive the blob (viettts_amd/dist.py) ----
    @property
    def packed_bytes(self) -> int:
        n = C.c_size_t(0)
        _lib.check(self.lib, self.lib.vtts_nat_duration_packed_bytes(self._h, C.byref(n)))
        return int(n.value)

    def packed_blob(self) -> torch.Tensor:
        if self._blob is None:
            raise RuntimeError("no parameters loaded")
        return self._blob

    def adopt_packed(self, blob: torch.Tensor) -> None:
        """Bind a packed blob produced by another rank's ``load_params`` (one broadcast at start-up, no data-path collective)."""
        if blob.dtype != torch.uint8 or blob.numel() < self.packed_bytes or blob.device != self.device:
            raise ValueError("packed blob must be a uint8 tensor of packed_bytes on this model's device")
        _lib.check(self.lib, self.lib.vtts_nat_duration_bind_packed(self._h, _ptr(blob), blob.numel()))
        self._blob = blob

    def load_params(self, params: HaikuDict, state: HaikuDict) -> None:
        """``dic["params"]`` and ``dic["aux"]`` of duration_latest_ckpt.pickle (text2mel.py:27-28)."""
        for mod, name, shape in self.param_table():
            src = state if name == "average" else params
            a = _lookup(src, mod, name)
            if a.shape != shape:
                raise ValueError(f"{mod}/{name}: checkpoint shape {a.shape}, the architecture needs {shape}")
            shp = (C.c_int64 * a.ndim)(*a.shape)
            _lib.check(self.lib, self.lib.vtts_nat_duration_set_param(self._h, mod.encode(), name.encode(), a.ctypes.data_as(C.c_void_p), shp, a.ndim))
        n = C.c_size_t(0)
        _lib.check(self.lib, self.lib.vtts_nat_duration_packed_bytes(self._h, C.byref(n)))
        blob = torch.empty(int(n.value), dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib, self.lib.vtts_nat_duration_pack(self._h, _ptr(blob), blob.numel(), C.c_void_p(stream.cuda_stream)))
        self._blob = blob

    def __call__(self, sentences: Sequence[Sequence[int]]) -> List[np.ndarray]:
        """Token-id lists -> per-sentence float32 arrays of seconds per token."""
        if len(sentences) == 0:
            return []
        out, lens = self.launch(sentences)
        host = out.cpu().numpy()
        return [host[i, : lens[i]].copy() for i in range(len(lens))]

    def launch(self, sentences: Sequence[Sequence[int]]):
        """The same forward pass, enqueued on the current stream and NOT waited for: ``(seconds [B, Lmax] on the device, lengths)``.  A pipeline
        records an event behind it, enqueues more work and reads the tensor back on a copy stream (viettts_amd/pipeline.py)."""
        if self._blob is None:
            raise RuntimeError("no parameters loaded")
        B = len(sentences)
        if B == 0:
            raise ValueError("empty batch")
        lens = [len(s) for s in sentences]
        if min(lens) < 1:
            raise ValueError("empty token sequence")
        Lmax = max(lens)
        tok = np.zeros((B, Lmax), dtype=np.int32)
        for i, s in enumerate(sentences):
            tok[i, : lens[i]] = np.asarray(s, dtype=np.int32)
        tok_d = torch.from_numpy(tok).to(self.device)
        len_d = torch.tensor(lens, dtype=torch.int32, device=self.device)
        out = torch.empty((B, Lmax), dtype=torch.float32, device=self.device)
        n = C.c_size_t(0)
        _lib.check(self.lib, self.lib.vtts_nat_duration_workspace_bytes(self._h, B, Lmax, C.byref(n)))
        if self._ws is None or self._ws.numel() < n.value:
            self._ws = torch.empty(int(n.value), dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(
                self.lib,
                self.lib.vtts_nat_duration_forward(self._h, _ptr(tok_d), _ptr(len_d), B, Lmax, _ptr(out), _ptr(self._ws), self._ws.numel(), C.c_void_p(stream.cuda_stream)),
            )
        return out, lens
