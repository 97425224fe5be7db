"""Host-side owner of the NAT acoustic model on one GPU.

Mirrors the network apply inside ``predict_mel(tokens, durations)`` of the reference (vietTTS/nat/text2mel.py:61-82):
``AcousticModel(is_training=False).inference(tokens, durations, n_frames)`` (vietTTS/nat/model.py:128-151) with the
checkpoint's ``params`` / ``aux`` dicts.  All arithmetic happens in the HIP library (include/vtts_nat.h); PyTorch-ROCm only
provides device memory and the stream.  There is no CPU path.

The prenet's dropout is on at inference in the reference (model.py:95-100) and draws from JAX's threefry PRNG through
Haiku's key chain, starting from the checkpoint's ``rng``.  ``dropout_rng`` (that key, uint32[2]) has the library draw THAT
stream on the GPU (jax.random's classic threefry layout; restated in oracle/nat_oracle.py, pinned by JAX's documented
``PRNGKey(0)`` answers, not by a JAX run); ``dropout_seeds`` draws this library's own per-sentence streams on the GPU (one
seed per sentence, independent of batching: the throughput pipeline's choice); ``keep_masks`` (boolean
``[n_frames, 2, 256]`` per sentence) makes the dropout explicit; none of them runs without dropout.
:func:`bernoulli_keep_masks` draws host masks from numpy's PCG64 for tests.
"""
from __future__ import annotations

import ctypes as C
import logging
import os
from typing import Optional, Sequence

import numpy as np
import torch

from .. import _lib
from .config import FLAGS
from .duration import HaikuDict, _lookup, _ptr


def bernoulli_keep_masks(n_frames: int, seed: int, prenet_dim: int = 256) -> np.ndarray:
    """``[n_frames, 2, prenet_dim]`` boolean keep masks, P(keep) = 0.5 (hk.dropout(key, 0.5, x), model.py:97, :99)."""
    return np.random.default_rng(seed).random((n_frames, 2, prenet_dim)) >= 0.5


_STREAM_LOGGED = set()


def _log_stream_once(partitionable: bool) -> None:
    """Which dropout mask stream the reference-faithful path draws (vietTTS/nat/text2mel.py:65-73 hands the checkpoint's rng to whatever JAX
    is installed; the reference pins no version): said once, so that a mel that differs from a JAX >= 0.5 run is not a silent surprise."""
    if partitionable not in _STREAM_LOGGED:
        _STREAM_LOGGED.add(partitionable)
        # WARNING, not INFO: invisible at the default log level, a mel that differs from a run of the reference under another JAX release
        # would be a silent surprise
        logging.getLogger("viettts_amd.nat").warning(
            "prenet dropout masks: jax.random threefry, %s layout, under dm-haiku's PRNGSequence key chain (set VTTS_JAX_THREEFRY_PARTITIONABLE=%d for the other one)",
            "partitionable (JAX >= 0.5 default; unpinned restatement)" if partitionable else "classic (JAX < 0.5 default)", 0 if partitionable else 1)


class AcousticModel:
    def __init__(self, device="cuda:0", lib_path=None, vocab_size: int = FLAGS.vocab_size, encoder_dim: int = FLAGS.acoustic_encoder_dim,
                 decoder_dim: int = FLAGS.acoustic_decoder_dim, prenet_dim: int = 256, mel_dim: int = FLAGS.mel_dim, postnet_dim: int = FLAGS.postnet_dim):
        self.lib = _lib.load(lib_path)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("AcousticModel needs a ROCm device ('cuda:N'); there is no CPU path")
        self.cfg = _lib.NatAcousticCfg(vocab_size, encoder_dim, decoder_dim, prenet_dim, mel_dim, postnet_dim)
        self.mel_dim, self.prenet_dim, self.encoder_dim = mel_dim, prenet_dim, encoder_dim
        self._h = C.c_void_p(0)
        dev_index = self.device.index if self.device.index is not None else 0
        _lib.check(self.lib, self.lib.vtts_nat_acoustic_create(C.byref(self.cfg), dev_index, C.byref(self._h)))
        self._blob: Optional[torch.Tensor] = None
        self._ws: Optional[torch.Tensor] = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.vtts_nat_acoustic_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def param_table(self):
        n = C.c_int(0)
        _lib.check(self.lib, self.lib.vtts_nat_acoustic_num_params(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            mod, name = C.c_char_p(), C.c_char_p()
            shape = (C.c_int64 * 3)()
            nd = C.c_int(0)
            _lib.check(self.lib, self.lib.vtts_nat_acoustic_param_info(self._h, i, C.byref(mod), C.byref(name), shape, C.byref(nd)))
            out.append((mod.value.decode(), name.value.decode(), tuple(int(shape[d]) for d in range(nd.value))))
        return out

    # ---- packed weights: rank 0 packs, the other ranks of a data-parallel job receive the blob (viettts_amd/dist.py) ----
    @property
    def packed_bytes(self) -> int:
        n = C.c_size_t(0)
        _lib.check(self.lib, self.lib.vtts_nat_acoustic_packed_bytes(self._h, C.byref(n)))
        return int(n.value)

    def packed_blob(self) -> torch.Tensor:
        if self._blob is None:
            raise RuntimeError("no parameters loaded")
        return self._blob

    def adopt_packed(self, blob: torch.Tensor) -> None:
        """Bind a packed blob produced by another rank's ``load_params`` (one broadcast at start-up, no data-path collective)."""
        if blob.dtype != torch.uint8 or blob.numel() < self.packed_bytes or blob.device != self.device:
            raise ValueError("packed blob must be a uint8 tensor of packed_bytes on this model's device")
        _lib.check(self.lib, self.lib.vtts_nat_acoustic_bind_packed(self._h, _ptr(blob), blob.numel()))
        self._blob = blob

    def load_params(self, params: HaikuDict, state: HaikuDict) -> None:
        """``dic["params"]`` and ``dic["aux"]`` of acoustic_latest_ckpt.pickle (text2mel.py:62-71)."""
        for mod, name, shape in self.param_table():
            a = _lookup(state if name == "average" else params, "acoustic_model/~/" + mod, name)
            if a.shape != shape:
                raise ValueError(f"{mod}/{name}: checkpoint shape {a.shape}, the architecture needs {shape}")
            shp = (C.c_int64 * a.ndim)(*a.shape)
            _lib.check(self.lib, self.lib.vtts_nat_acoustic_set_param(self._h, mod.encode(), name.encode(), a.ctypes.data_as(C.c_void_p), shp, a.ndim))
        n = C.c_size_t(0)
        _lib.check(self.lib, self.lib.vtts_nat_acoustic_packed_bytes(self._h, C.byref(n)))
        blob = torch.empty(int(n.value), dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib, self.lib.vtts_nat_acoustic_pack(self._h, _ptr(blob), blob.numel(), C.c_void_p(stream.cuda_stream)))
        self._blob = blob

    def device_keep_masks(self, seeds: Sequence[int], Fmax: int) -> torch.Tensor:
        """``[B, Fmax, 2, prenet_dim]`` uint8 keep masks drawn on the GPU (include/vtts_nat.h: Threefry-2x32-20, one
        64-bit seed per sentence; oracle/nat_oracle.py::threefry_keep_masks is the CPU restatement)."""
        B = len(seeds)
        sd = torch.tensor([int(x) & 0x7FFFFFFFFFFFFFFF for x in seeds], dtype=torch.int64, device=self.device)
        keep = torch.empty((B, int(Fmax), 2, self.prenet_dim), dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib, self.lib.vtts_nat_acoustic_keep_masks(self._h, _ptr(sd), B, int(Fmax), _ptr(keep), C.c_void_p(stream.cuda_stream)))
        return keep

    def device_keep_masks_haiku(self, rng_key, B: int, Fmax: int, partitionable: Optional[bool] = None) -> torch.Tensor:
        """``[B, Fmax, 2, prenet_dim]`` uint8 keep masks as the REFERENCE draws them from the checkpoint's ``rng`` (a
        jax.random.PRNGKey, uint32[2]): jax.random's threefry under dm-haiku's key chain (include/vtts_nat.h:
        vtts_nat_acoustic_keep_masks_haiku[_mode]; restated in oracle/nat_oracle.py::haiku_prenet_keep_masks).  The same masks for
        every sentence, as every run of the reference starts from the same key.

        ``partitionable``: which counter layout — False = the classic one (every JAX before 0.5, pinned by JAX's documented
        known answers), True = ``jax_threefry_partitionable`` (JAX >= 0.5's default; restated from recollection, unpinned).  None
        reads ``VTTS_JAX_THREEFRY_PARTITIONABLE`` (unset / 0 = classic).  The stream in use is logged once per process."""
        if partitionable is None:
            partitionable = os.environ.get("VTTS_JAX_THREEFRY_PARTITIONABLE", "0") not in ("", "0")
        _log_stream_once(bool(partitionable))
        k = np.asarray(rng_key, dtype=np.uint32).reshape(2)
        keep = torch.empty((int(B), int(Fmax), 2, self.prenet_dim), dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib, self.lib.vtts_nat_acoustic_keep_masks_haiku_mode(self._h, int(k[0]), int(k[1]), int(bool(partitionable)), int(B), int(Fmax),
                                                                                  _ptr(keep), C.c_void_p(stream.cuda_stream)))
        return keep

    def set_option(self, key: str, value: int) -> None:
        """``"bf16x3"``: 1 = the postnet's matrix products as three bf16 x bf16 terms on the bf16 matrix pipe (include/vtts_nat.h); 0 = fp32 (default)."""
        _lib.check(self.lib, self.lib.vtts_nat_acoustic_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key: str) -> int:
        v = C.c_int(0)
        _lib.check(self.lib, self.lib.vtts_nat_acoustic_get_option(self._h, key.encode(), C.byref(v)))
        return int(v.value)

    def wait_group(self, group: int, stream: Optional[torch.cuda.Stream] = None) -> None:
        """Make ``stream`` (default: torch's current stream) wait until the rows of ``group`` of the last ``group_row0`` call are complete."""
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib, self.lib.vtts_nat_acoustic_wait_group(self._h, int(group), C.c_void_p(st.cuda_stream)))

    def encode(self, sentences: Sequence[Sequence[int]]) -> torch.Tensor:
        """The token encoder alone (include/vtts_nat.h: vtts_nat_acoustic_encode), enqueued on the current stream: ``[B, Lmax, 2 * encoder_dim]`` on the
        device.  It needs the tokens only — a pipeline runs it while the host still turns durations into frame counts — and a row does not depend
        on its batch: select / re-order rows (``enc[rows]``) and hand them to :meth:`__call__` as ``encoded``."""
        if self._blob is None:
            raise RuntimeError("no parameters loaded")
        B = len(sentences)
        lens = [len(s) for s in sentences]
        if B == 0 or min(lens) < 1:
            raise ValueError("empty batch or token sequence")
        Lmax = max(lens)
        tok = np.zeros((B, Lmax), dtype=np.int32)
        for i, s in enumerate(sentences):
            tok[i, : lens[i]] = np.asarray(s, dtype=np.int32)
        tok_d = torch.from_numpy(tok).to(self.device)
        len_d = torch.tensor(lens, dtype=torch.int32, device=self.device)
        enc = torch.empty((B, Lmax, 2 * self.encoder_dim), dtype=torch.float32, device=self.device)
        n = C.c_size_t(0)
        _lib.check(self.lib, self.lib.vtts_nat_acoustic_workspace_bytes(self._h, B, Lmax, 1, C.byref(n)))
        if self._ws is None or self._ws.numel() < n.value:
            self._ws = torch.empty(int(n.value), dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib, self.lib.vtts_nat_acoustic_encode(self._h, _ptr(tok_d), _ptr(len_d), B, Lmax, _ptr(enc), _ptr(self._ws), self._ws.numel(),
                                                                   C.c_void_p(stream.cuda_stream)))
        return enc

    def __call__(self, sentences: Sequence[Sequence[int]], durations_frames: Sequence[np.ndarray], n_frames: Sequence[int],
                 keep_masks: Optional[Sequence[np.ndarray]] = None, dropout_seeds: Optional[Sequence[int]] = None, to_host: bool = True,
                 dropout_rng=None, group_row0: Optional[Sequence[int]] = None, encoded: Optional[torch.Tensor] = None):
        """Per sentence: token ids, per-token durations in FRAMES, number of frames -> mel ``[n_frames, mel_dim]``.
        Dropout: explicit ``keep_masks`` (host arrays), or ``dropout_rng`` (the checkpoint's jax PRNGKey, uint32[2]: the
        reference's own mask stream, drawn on the GPU), or ``dropout_seeds`` (one int per sentence: this library's own
        per-sentence streams, drawn on the GPU), or none of them (no dropout).  ``to_host=False`` returns the device tensor ``[B, Fmax, mel_dim]`` (rows past a
        sentence's ``n_frames`` are zero) instead of per-sentence host arrays: the generator's input stays in HBM.
        ``group_row0`` (with ``to_host=False``): row boundaries ``[0, ..., B]`` of groups whose mel is handed over as soon as the decoder has
        produced the group's last frame (include/vtts_nat.h: vtts_nat_acoustic_forward_groups); a consumer stream waits for group g with
        :meth:`wait_group`.  Same mel, bit for bit.
        ``encoded``: these sentences' rows of an :meth:`encode` output, in this call's order (``[B, L, 2 * encoder_dim]``, L >= the longest
        sentence): the token encoder is skipped.  Same mel, bit for bit."""
        if self._blob is None:
            raise RuntimeError("no parameters loaded")
        B = len(sentences)
        lens = [len(s) for s in sentences]
        if B == 0 or min(lens) < 1 or min(n_frames) < 1:
            raise ValueError("empty batch, token sequence or frame count")
        Lmax, Fmax = max(lens), int(max(n_frames))
        # every argument check before anything is uploaded or grown
        if group_row0 is not None and to_host:
            raise ValueError("group_row0 hands the mel over on the device: use to_host=False")
        if dropout_seeds is not None and keep_masks is None and dropout_rng is None and len(dropout_seeds) != B:
            raise ValueError("one dropout seed per sentence")
        if encoded is not None:
            if encoded.dim() != 3 or encoded.shape[0] != B or encoded.shape[1] < Lmax or encoded.shape[2] != 2 * self.encoder_dim or encoded.dtype != torch.float32:
                raise ValueError(f"encoded must be float32 [{B}, >= {Lmax}, {2 * self.encoder_dim}] (got {tuple(encoded.shape)} {encoded.dtype})")
            encoded = encoded.contiguous()
            Lmax = int(encoded.shape[1])
        tok = np.zeros((B, Lmax), dtype=np.int32) if encoded is None else None  # (the C side ignores the tokens when the encoder's output is given)
        dur = np.zeros((B, Lmax), dtype=np.float32)
        for i, s in enumerate(sentences):
            if tok is not None:
                tok[i, : lens[i]] = np.asarray(s, dtype=np.int32)
            dur[i, : lens[i]] = np.asarray(durations_frames[i], dtype=np.float32).reshape(-1)
        keep_d = None
        if keep_masks is not None:
            keep = np.zeros((B, Fmax, 2, self.prenet_dim), dtype=np.uint8)
            for i, m in enumerate(keep_masks):
                keep[i, : n_frames[i]] = np.asarray(m, dtype=bool)[: n_frames[i]]
            keep_d = torch.from_numpy(keep).to(self.device)
        elif dropout_rng is not None:
            keep_d = self.device_keep_masks_haiku(dropout_rng, B, Fmax)
        elif dropout_seeds is not None:
            keep_d = self.device_keep_masks(dropout_seeds, Fmax)
        tok_d = torch.from_numpy(tok).to(self.device) if tok is not None else None
        dur_d = torch.from_numpy(dur).to(self.device)
        len_d = torch.tensor(lens, dtype=torch.int32, device=self.device)
        nf_d = torch.tensor([int(n) for n in n_frames], dtype=torch.int32, device=self.device)
        out = torch.empty((B, Fmax, self.mel_dim), dtype=torch.float32, device=self.device)
        n = C.c_size_t(0)
        _lib.check(self.lib, self.lib.vtts_nat_acoustic_workspace_bytes(self._h, B, Lmax, Fmax, C.byref(n)))
        if self._ws is None or self._ws.numel() < n.value:
            self._ws = torch.empty(int(n.value), dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            if encoded is not None:
                r0 = [int(v) for v in group_row0] if group_row0 is not None else [0]
                ng = len(r0) - 1
                gfr = [max(int(n) for n in n_frames[r0[g] : r0[g + 1]]) if r0[g + 1] > r0[g] else 0 for g in range(ng)]
                _lib.check(
                    self.lib,
                    self.lib.vtts_nat_acoustic_forward_from_encoder(self._h, _ptr(encoded), _ptr(len_d), _ptr(dur_d), _ptr(nf_d), B, Lmax, Fmax, _ptr(keep_d),
                                                                    _ptr(out), _ptr(self._ws), self._ws.numel(), C.c_void_p(stream.cuda_stream), ng,
                                                                    (C.c_int32 * (ng + 1))(*r0), (C.c_int32 * max(ng, 1))(*(gfr or [0]))),
                )
            elif group_row0 is not None:
                r0 = [int(v) for v in group_row0]
                ng = len(r0) - 1
                gfr = [max(int(n) for n in n_frames[r0[g] : r0[g + 1]]) if r0[g + 1] > r0[g] else 0 for g in range(ng)]
                _lib.check(
                    self.lib,
                    self.lib.vtts_nat_acoustic_forward_groups(self._h, _ptr(tok_d), _ptr(len_d), _ptr(dur_d), _ptr(nf_d), B, Lmax, Fmax, _ptr(keep_d), _ptr(out),
                                                              _ptr(self._ws), self._ws.numel(), C.c_void_p(stream.cuda_stream), ng,
                                                              (C.c_int32 * (ng + 1))(*r0), (C.c_int32 * ng)(*gfr)),
                )
            else:
                _lib.check(
                    self.lib,
                    self.lib.vtts_nat_acoustic_forward(self._h, _ptr(tok_d), _ptr(len_d), _ptr(dur_d), _ptr(nf_d), B, Lmax, Fmax, _ptr(keep_d), _ptr(out),
                                                       _ptr(self._ws), self._ws.numel(), C.c_void_p(stream.cuda_stream)),
                )
        if not to_host:
            return out
        host = out.cpu().numpy()
        return [host[i, : n_frames[i]].copy() for i in range(B)]
