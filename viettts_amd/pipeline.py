"""Text -> waveform for many sentences on one rank of a data-parallel job (BASELINE.json configs[3]).

The reference synthesises one sentence per process invocation (vietTTS/synthesizer.py:33-39: text2mel, then mel2wave).
Sentences are independent at every step, so a corpus shards with no exchange step: every rank takes the sentences
``shard_utterances`` deals it (by token count — known before any network runs), runs the three models on its own GPU
and keeps its waveforms.  The only collectives of the whole job are the three start-up weight broadcasts
(viettts_amd/dist.py).  Within a rank the stages run batched:
    tokens --DurationModel--> seconds/token --rules (text2mel.py:90-97)--> frames --AcousticModel--> mel --Generator--> wav
with the generator fed length buckets (equal-length mels batch without padding: the generator has no masking, and a
padded batch would change nothing for the true frames but costs time).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .dist import shard_utterances
from .nat import text2mel as t2m
from .nat.config import FLAGS


def synthesize_sentences(token_lists: Sequence[Sequence[int]], duration_model, acoustic_model, generator, silence_duration: float = -1.0,
                         dropout_seed: Optional[int] = 0, rank: int = 0, world: int = 1) -> Dict[int, np.ndarray]:
    """Waveforms (float32, 16 kHz samples) of THIS rank's sentences, keyed by sentence index."""
    mine = shard_utterances([len(t) for t in token_lists], world)[rank]
    if not mine:
        return {}
    toks = [list(token_lists[i]) for i in mine]
    secs = duration_model(toks)  # [L] seconds per token each
    frames, nfr, trail = [], [], []
    for t, d in zip(toks, secs):
        d = t2m.apply_duration_rules(t, d[None, :], silence_duration)  # text2mel.py:90-97
        frames.append(t2m.durations_to_frames(d)[0])  # :78
        nfr.append(t2m.n_frames_from_durations(d))  # :79
        trail.append(t2m.trailing_silence_frames(d) if t[-1] == FLAGS.sil_index else 0)  # :99-101
    keep = None
    if dropout_seed is not None:
        from .nat.acoustic import bernoulli_keep_masks

        keep = [bernoulli_keep_masks(max(n, 1), dropout_seed + i) for i, n in zip(mine, nfr)]
    ok = [k for k, n in enumerate(nfr) if n >= 1]
    mels: List[Optional[np.ndarray]] = [None] * len(mine)
    if ok:
        out = acoustic_model([toks[k] for k in ok], [frames[k] for k in ok], [nfr[k] for k in ok],
                             keep_masks=None if keep is None else [keep[k] for k in ok])
        for k, m in zip(ok, out):
            mels[k] = m[: m.shape[0] - trail[k]] if trail[k] else m  # :102
    # length buckets through the generator
    by_len: Dict[int, List[int]] = {}
    for k, m in enumerate(mels):
        if m is not None and m.shape[0] > 0:
            by_len.setdefault(m.shape[0], []).append(k)
    wavs: Dict[int, np.ndarray] = {}
    for T, ks in by_len.items():
        batch = torch.from_numpy(np.stack([mels[k] for k in ks])).to(generator.device)
        w = generator(batch).cpu().numpy()
        for r, k in enumerate(ks):
            wavs[mine[k]] = w[r]
    for k, m in enumerate(mels):
        if mine[k] not in wavs:
            wavs[mine[k]] = np.zeros((0,), np.float32)
    return wavs
