"""Text -> waveform for many sentences on one rank of a data-parallel job (BASELINE.json configs[3]).

The reference synthesises one sentence per process invocation (vietTTS/synthesizer.py:33-39: text2mel, then mel2wave).
Sentences are independent at every step, so a corpus shards with no exchange step: every rank takes the sentences
``shard_utterances`` deals it (by token count — known before any network runs), runs the three models on its own GPU
and keeps its waveforms.  The only collectives of the whole job are the three start-up weight broadcasts
(viettts_amd/dist.py).  Within a rank the stages run batched:
    tokens --DurationModel--> seconds/token --rules (text2mel.py:90-97)--> frames --AcousticModel--> mel --Generator--> wav
with the generator fed ragged batches (sentences sorted by length, each batch padded to its longest; the bf16 engine
gives every utterance the zero padding it would see alone and skips the tiles past its end).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .dist import shard_utterances
from .nat import text2mel as t2m
from .nat.config import FLAGS


def synthesize_sentences(token_lists: Sequence[Sequence[int]], duration_model, acoustic_model, generator, silence_duration: float = -1.0,
                         dropout_seed: Optional[int] = 0, rank: int = 0, world: int = 1, gen_batch: int = 64) -> Dict[int, np.ndarray]:
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
    ok = [k for k, n in enumerate(nfr) if n >= 1]
    mels: List[Optional[np.ndarray]] = [None] * len(mine)
    if ok:
        # prenet dropout (on at inference, model.py:95-100): masks drawn on the GPU, seeded by the sentence's GLOBAL index
        out = acoustic_model([toks[k] for k in ok], [frames[k] for k in ok], [nfr[k] for k in ok],
                             dropout_seeds=None if dropout_seed is None else [dropout_seed + mine[k] for k in ok])
        for k, m in zip(ok, out):
            mels[k] = m[: m.shape[0] - trail[k]] if trail[k] else m  # :102
    # the generator takes ragged batches (vtts_hifigan_forward_ragged: each utterance's samples are those of running it
    # alone): sentences sorted by length, dealt into batches of at most `gen_batch`, padded to the batch's longest
    todo = sorted((k for k, m in enumerate(mels) if m is not None and m.shape[0] > 0), key=lambda k: mels[k].shape[0])
    wavs: Dict[int, np.ndarray] = {}
    ragged = getattr(generator, "dtype_name", "") == "bf16"  # the fp32 engine takes one utterance (length) at a time
    for i0 in range(0, len(todo), gen_batch if ragged else 1):
        ks = todo[i0 : i0 + (gen_batch if ragged else 1)]
        fr = [mels[k].shape[0] for k in ks]
        batch = np.zeros((len(ks), max(fr), mels[ks[0]].shape[1]), dtype=np.float32)
        for r, k in enumerate(ks):
            batch[r, : fr[r]] = mels[k]
        dev = torch.from_numpy(batch).to(generator.device)
        w = (generator.forward_ragged(dev, fr) if ragged else generator(dev)).cpu().numpy()
        for r, k in enumerate(ks):
            wavs[mine[k]] = w[r, : generator.hop * fr[r]].copy()
    for k, m in enumerate(mels):
        if mine[k] not in wavs:
            wavs[mine[k]] = np.zeros((0,), np.float32)
    return wavs
