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

from typing import List, Dict, Optional, Sequence

import numpy as np
import torch

from .dist import shard_utterances
from .longform import _pass_frames
from .nat import text2mel as t2m


PASS_FRAMES = 65536  # fallback for generators without the "pass_frames" option; mel frames per pass the generator's launches are sized for (engine.hip: pick_microbatch)


def _generator_batches(rows: Sequence[int], frames: Sequence[int], gen_batch: int = 0, pass_frames: int = PASS_FRAMES) -> List[List[int]]:
    """Cut ``rows`` (sorted by ascending ``frames``) into the generator's ragged batches.  gen_batch > 0: at most that many sentences
    per batch.  gen_batch = 0: as few passes as PASS_FRAMES of REAL frames each allow (the ragged kernels skip the tiles past an
    utterance's end, so padding costs workspace, not time), the passes balanced by frames: 256 transcript sentences (54.7k frames,
    longest 281) are ONE pass instead of four of 64 sentences — fewer, larger launches (generator stage 38.1 -> 34 ms)."""
    rows = list(rows)
    if not rows:
        return []
    if gen_batch > 0:
        return [rows[i : i + gen_batch] for i in range(0, len(rows), gen_batch)]
    total = int(sum(frames))
    k = max(1, -(-total // int(pass_frames * 1.25)))
    out, acc, cut = [[]], 0, 1
    for r, f in zip(rows, frames):
        if acc >= total * cut / k and cut < k and out[-1]:
            out.append([])
            cut += 1
        out[-1].append(r)
        acc += int(f)
    return out


def synthesize_sentences(token_lists: Sequence[Sequence[int]], duration_model, acoustic_model, generator, silence_duration: float = -1.0,
                         dropout_seed: Optional[int] = 0, rank: int = 0, world: int = 1, gen_batch: int = 0,
                         timing: Optional[dict] = None) -> Dict[int, np.ndarray]:
    """Waveforms (float32, 16 kHz samples) of THIS rank's sentences, keyed by sentence index.  ``timing`` (a dict) receives
    device-synchronised wall seconds per stage."""
    import time

    def mark(name, t_prev):
        if timing is None:
            return t_prev
        torch.cuda.synchronize()
        now = time.perf_counter()
        timing[name] = timing.get(name, 0.0) + now - t_prev
        return now

    mine = shard_utterances([len(t) for t in token_lists], world)[rank]
    if not mine:
        return {}
    t_last = 0.0
    if timing is not None:
        torch.cuda.synchronize()
        t_last = time.perf_counter()
    toks = [list(token_lists[i]) for i in mine]
    secs = duration_model(toks)  # [L] seconds per token each
    t_last = mark("duration_s", t_last)
    frames, nfr, trail = t2m.frame_plan(toks, secs, silence_duration)  # text2mel.py:78-79, :90-102 for the whole shard at once
    # longest first: the decoder steps all sentences together and a 64-sentence tile leaves the per-frame launches once ITS longest
    # sentence is done, so tiles of similar lengths finish early (a sentence's mel does not depend on its row: rows are independent)
    ok = sorted((k for k, n in enumerate(nfr) if n >= 1), key=lambda k: (-nfr[k], k))
    t_last = mark("host_rules_s", t_last)
    wavs: Dict[int, np.ndarray] = {}
    gfr = {k: nfr[k] - trail[k] for k in ok}  # frames the generator sees: the mel minus its trailing silence (:102)
    if ok:
        # prenet dropout (on at inference, model.py:95-100): masks drawn on the GPU, seeded by the sentence's GLOBAL index.
        # The mel stays in HBM: [len(ok), Fmax, 80] on the device, rows past a sentence's frames zero.
        mel_dev = acoustic_model([toks[k] for k in ok], [frames[k] for k in ok], [nfr[k] for k in ok],
                                 dropout_seeds=None if dropout_seed is None else [dropout_seed + mine[k] for k in ok], to_host=False)
        t_last = mark("acoustic_s", t_last)
        # the generator takes ragged batches (vtts_hifigan_forward_ragged: each utterance's samples are those of running it
        # alone): sentences sorted by length, dealt into batches of at most `gen_batch`, cut to the batch's longest
        ragged = getattr(generator, "dtype_name", "") == "bf16"  # the fp32 engine takes one utterance (length) at a time
        todo = sorted((r for r, k in enumerate(ok) if gfr[k] > 0), key=lambda r: gfr[ok[r]])
        pending = []
        for rows in _generator_batches(todo, [gfr[ok[r]] for r in todo], gen_batch if ragged else 1, _pass_frames(generator)):
            fr = [gfr[ok[r]] for r in rows]
            batch = mel_dev[torch.tensor(rows, device=mel_dev.device), : max(fr)].contiguous()  # a device-side gather (plumbing)
            if ragged:
                prev = generator.get_option("microbatch")
                if prev == 0:
                    generator.set_option("microbatch", len(rows))  # one pass for the batch (the engine's own cut is PASS_FRAMES of PADDED frames)
                try:
                    w = generator.forward_ragged(batch, fr)
                finally:
                    generator.set_option("microbatch", prev)  # a caller's own setting is left alone
            else:
                w = generator(batch)
            host = torch.empty(w.shape, dtype=w.dtype, pin_memory=True)
            host.copy_(w, non_blocking=True)  # pinned, stream-ordered: the next batch computes behind this copy's enqueue
            pending.append((rows, fr, host))
        torch.cuda.synchronize()
        for rows, fr, host in pending:
            hn = host.numpy()
            for q, r in enumerate(rows):
                wavs[mine[ok[r]]] = hn[q, : generator.hop * fr[q]]  # a view of the batch's pinned buffer (kept alive by the view)
    else:
        t_last = mark("acoustic_s", t_last)
    t_last = mark("generator_s", t_last)
    if timing is not None:
        timing["frames"] = int(sum(gfr.values()))
        timing["frames_max"] = int(max(list(gfr.values()) or [0]))
        timing["tokens"] = int(sum(len(t) for t in toks))
    for k in range(len(mine)):
        if mine[k] not in wavs:
            wavs[mine[k]] = np.zeros((0,), np.float32)
    return wavs
