"""Long-form synthesis by exact chunking (BASELINE.json configs[4]; new capability — the reference's
``mel2wave`` processes a whole utterance in one shot, vietTTS/hifigan/mel2wave.py:37-38).

The generator's receptive field is finite: an output sample depends on mel frames within +-12.71
frames.  A chunk fed with a 13-frame halo per side therefore reproduces the
un-chunked samples of its interior up to fp32 reassociation — no cross-fade, no window.  At true
utterance edges no halo is added, so the generator's own zero padding applies exactly as un-chunked.
Chunks are independent: they batch on one GPU and shard chunk c -> rank c mod world across GPUs with
no data-path collective (viettts_amd/dist.py).
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Sequence

import torch

from .dist import HALO_FRAMES, Chunk, plan_chunks, shard_chunks


PASS_FRAMES = 65536  # fallback for generators without the "pass_frames" option; mel frames per pass through the generator the engines are sized for (engine.hip: pick_microbatch)


def _pass_frames(gen) -> int:
    """The engine's own figure (engine.hip: pass_frames) where the generator exposes it."""
    try:
        return int(gen.get_option("pass_frames"))
    except Exception:
        return PASS_FRAMES


def _run_group(gen, mel: torch.Tensor, chunks: Sequence[Chunk], out: torch.Tensor, max_batch: int) -> None:
    """All chunks in `chunks` have the same fed length; run them as batches and scatter the kept
    samples into `out` ([hop*T]).  max_batch = 0: as many chunks per pass as make one full-size pass
    (a 10-minute utterance in 512-frame chunks: all 73 after the first; 28.6 -> 26.3 ms against 16 per pass)."""
    hop = gen.hop
    if max_batch <= 0 and chunks:
        max_batch = max(1, _pass_frames(gen) // max(1, chunks[0].hi - chunks[0].lo))
    for i in range(0, len(chunks), max_batch):
        grp = chunks[i : i + max_batch]
        batch = torch.stack([mel[c.lo : c.hi] for c in grp]).contiguous()
        wav = gen(batch)
        n, c0 = len(grp), grp[0]
        keep = c0.t1 - c0.t0
        if all(c.keep_from == c0.keep_from and c.t1 - c.t0 == keep and c.t0 == c0.t0 + r * keep for r, c in enumerate(grp)):
            # consecutive chunks with the same geometry (every interior chunk of an utterance): their kept samples are one contiguous range
            out[hop * c0.t0 : hop * (c0.t0 + n * keep)].view(n, hop * keep).copy_(wav[:, hop * c0.keep_from : hop * (c0.keep_from + keep)])
            continue
        for r, c in enumerate(grp):
            out[hop * c.t0 : hop * c.t1] = wav[r, hop * c.keep_from : hop * (c.keep_from + c.t1 - c.t0)]


def synthesize_chunked(gen, mel: torch.Tensor, chunk_frames: int = 512, halo: int = HALO_FRAMES, max_batch: int = 0,
                       rank: int = 0, world: int = 1, timing: Optional[Dict] = None) -> torch.Tensor:
    """mel ``[T, num_mels]`` float32 on the generator's device -> ``[hop*T]`` float32 on the same
    device.  With ``world > 1`` only this rank's chunks (c mod world == rank) are computed; the other
    ranges of the result are left zero (the caller gathers or writes per-rank pieces).
    ``timing``, if a dict, receives ``first_chunk_s`` (time to the first chunk's samples) and ``total_s``."""
    if mel.dim() != 2:
        raise ValueError("mel must be [T, num_mels]")
    T = mel.shape[0]
    chunks = shard_chunks(plan_chunks(T, chunk_frames, halo), world)[rank]
    out = torch.zeros(gen.hop * T, dtype=torch.float32, device=mel.device)
    t_start = time.perf_counter()
    # the first chunk alone first: streaming latency = time-to-first-chunk
    if chunks:
        _run_group(gen, mel, chunks[:1], out, 1)
        if timing is not None:
            torch.cuda.synchronize(mel.device)
            timing["first_chunk_s"] = time.perf_counter() - t_start
    rest = chunks[1:]
    by_len: Dict[int, List[Chunk]] = {}
    for c in rest:
        by_len.setdefault(c.frames, []).append(c)
    for _, grp in sorted(by_len.items(), key=lambda kv: -kv[0]):
        _run_group(gen, mel, grp, out, max_batch)
    if timing is not None:
        torch.cuda.synchronize(mel.device)
        timing["total_s"] = time.perf_counter() - t_start
        timing["chunks"] = len(chunks)
    return out
