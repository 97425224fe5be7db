"""Text -> waveform for many sentences on one rank of a data-parallel job (BASELINE.json configs[3]).

The reference synthesises one sentence per process invocation (vietTTS/synthesizer.py:33-39: text2mel, then mel2wave).
Sentences are independent at every step, so a corpus shards with no exchange step: every rank takes the sentences
``shard_utterances`` deals it (by token count — known before any network runs), runs the three models on its own GPU
and keeps its waveforms.  The only collectives of the whole job are the three start-up weight broadcasts
(viettts_amd/dist.py).  Within a rank the stages run batched:
    tokens --DurationModel--> seconds/token --rules (text2mel.py:90-97)--> frames --AcousticModel--> mel --Generator--> wav
with the generator fed ragged batches (sentences sorted by length, each batch padded to its longest; every engine — bf16, and since
round 5 fp32 and bf16x3, the ones that answer to the reference's 1e-4 — gives every utterance the zero padding it would see alone and
skips the tiles past its end).  (Round 4 also carried an opt-in schedule that overlapped the last two stages on streams of
their own; it measured 3 % faster or 30 % slower depending on how the runtime mapped streams to hardware queues — 68.0 against 51.2 ms on
the round's driver box — and was deleted in round 5: profiles/r04_c_kernel_structure_findings.md)
"""
from __future__ import annotations

import os
from typing import List, Dict, Optional, Sequence

import numpy as np
import torch

from .dist import shard_utterances
from .longform import _pass_frames
from .nat import text2mel as t2m


PASS_FRAMES = 65536  # fallback for generators without the "pass_frames" option; mel frames per pass the generator's launches are sized for (engine.hip: pick_microbatch)


SPLIT_MIN_FRAMES = 16384  # below this a second pass costs more than the read-back it hides
TAIL_SHARE = 0.25  # measured 0.5 / 0.35 / 0.25: 52.7 / 53.0 / 52.3 ms (tools/experiments/r04/run29.sh)


def _generator_batches(rows: Sequence[int], frames: Sequence[int], gen_batch: int = 0, pass_frames: int = PASS_FRAMES) -> List[List[int]]:
    """Cut ``rows`` (sorted by ascending ``frames``) into the generator's ragged batches.  gen_batch > 0: at most that many sentences
    per batch.  gen_batch = 0: as few passes as PASS_FRAMES of REAL frames each allow (the ragged kernels skip the tiles past an
    utterance's end, so padding costs workspace, not time), the passes balanced by frames: 256 transcript sentences (54.7k frames,
    longest 281) are ONE pass instead of four of 64 sentences — fewer, larger launches (generator stage 38.1 -> 34 ms).
    Padding does cost WORKSPACE (4 buffers x 8192 x 2 B per padded frame): a batch is also closed before its padded size,
    ``len(batch) x longest``, would pass 2 x PASS_FRAMES — 500 sentences of 100 frames and one of 1000 are two passes, not one of
    500 000 padded frames (32 GB)."""
    rows = list(rows)
    if not rows:
        return []
    if gen_batch > 0:
        return [rows[i : i + gen_batch] for i in range(0, len(rows), gen_batch)]
    total = int(sum(frames))
    k = max(1, -(-total // int(pass_frames * 1.25)))
    out, acc, cut = [[]], 0, 1
    for r, f in zip(rows, frames):
        if out[-1] and ((acc >= total * cut / k and cut < k) or (len(out[-1]) + 1) * int(f) > 2 * pass_frames):
            if acc >= total * cut / k and cut < k:
                cut += 1
            out.append([])
        out[-1].append(r)
        acc += int(f)
    # A job that fits ONE pass still leaves in two when it is large: a pass's waveforms go to the host on a copy stream while the next pass
    # computes, and the only pass's read-back (73 MB for the 256 transcript sentences) would be exposed in full.  The last pass — the longest
    # sentences — takes TAIL_SHARE of the frames: its read-back is what stays exposed (256 sentences: 53.5 -> 52.3 ms).
    if len(out) == 1 and len(rows) >= 32 and total >= SPLIT_MIN_FRAMES:
        acc, cut = 0, len(rows) - 1
        for i, f in enumerate(frames):
            acc += int(f)
            if acc >= total * (1.0 - TAIL_SHARE):
                cut = i + 1
                break
        cut = min(max(cut, 1), len(rows) - 1)
        out = [rows[:cut], rows[cut:]]
    return out


_SIDE_STREAMS: dict = {}


def _copy_stream(device: torch.device):
    """The stream a pass's waveforms leave on (pinned host memory) while the next pass computes; one per device, created once."""
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def synthesize_sentences(token_lists: Sequence[Sequence[int]], duration_model, acoustic_model, generator, silence_duration: float = -1.0,
                         dropout_seed: Optional[int] = 0, rank: int = 0, world: int = 1, gen_batch: int = 0,
                         timing: Optional[dict] = None) -> Dict[int, np.ndarray]:
    """Waveforms (float32, 16 kHz samples) of THIS rank's sentences, keyed by sentence index.  ``timing`` (a dict) receives
    wall seconds per stage.  The stages run one after the other on the caller's stream; a pass's waveforms leave for pinned host memory
    on a copy stream while the next pass computes.  Masks are seeded by the GLOBAL sentence index and every stage computes a row
    independently of its batch, so the samples depend neither on the shard nor on the batch (tests/test_gpu_nat.py: bit-identical to each
    sentence alone on the bf16 and bf16x3 engines; ~1e-7 on f32, whose transposed convolution picks its kernel — and with it a summation
    order — by the pass's slot length).  ``generator`` may be any engine: bf16 (throughput; ~1e-2), bf16x3 or f32 (the reference's 1e-4)."""
    import time

    def mark(name, t_prev, sync=True):
        if timing is None:
            return t_prev
        if sync:
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
    enc_all = None
    if hasattr(duration_model, "launch") and hasattr(acoustic_model, "encode") and torch.cuda.is_available() and not os.environ.get("VTTS_PIPE_NO_EARLY_ENCODE"):
        # The acoustic model's token encoder needs the tokens only: it is enqueued right behind the duration model and runs while the host reads the
        # durations back (on the copy stream, behind an event) and turns them into frame counts — the GPU idled through that (~1 ms per 256 sentences).
        # A row of the encoder's output does not depend on its batch (include/vtts_nat.h: vtts_nat_acoustic_encode): the rows are picked in the
        # acoustic call's order below.
        dev0 = getattr(acoustic_model, "device", None)
        cur0 = torch.cuda.current_stream(dev0)
        sec_dev, lens0 = duration_model.launch(toks)
        ev_dur = torch.cuda.Event()
        ev_dur.record(cur0)
        enc_all = acoustic_model.encode(toks)
        s_cp = _copy_stream(torch.device(dev0) if not isinstance(dev0, torch.device) else dev0)
        s_cp.wait_event(ev_dur)
        with torch.cuda.stream(s_cp):
            sec_host = torch.empty(sec_dev.shape, dtype=sec_dev.dtype, pin_memory=True)
            sec_host.copy_(sec_dev, non_blocking=True)
            ev_host = torch.cuda.Event()
            ev_host.record(s_cp)
        sec_dev.record_stream(s_cp)
        ev_host.synchronize()
        hn = sec_host.numpy()
        secs = [hn[i, : lens0[i]].copy() for i in range(len(lens0))]
        t_last = mark("duration_s", t_last, sync=False)  # (the encoder is still running: its time shows up in acoustic_s)
    else:
        secs = duration_model(toks)  # [L] seconds per token each
        t_last = mark("duration_s", t_last)
    frames, nfr, trail = t2m.frame_plan(toks, secs, silence_duration)  # text2mel.py:78-79, :90-102 for the whole shard at once
    # longest first: the decoder steps all sentences together and a 64-sentence tile leaves the per-frame launches once ITS longest
    # sentence is done, so tiles of similar lengths finish early (a sentence's mel does not depend on its row: rows are independent)
    ok = sorted((k for k, n in enumerate(nfr) if n >= 1), key=lambda k: (-nfr[k], k))
    t_last = mark("host_rules_s", t_last, sync=enc_all is None)
    wavs: Dict[int, np.ndarray] = {}
    gfr = {k: nfr[k] - trail[k] for k in ok}  # frames the generator sees: the mel minus its trailing silence (:102)
    if ok:
        dev = generator.device
        ragged = hasattr(generator, "forward_ragged")  # every engine takes ragged batches (round 5: fp32 and bf16x3 too — the parity-grade pipeline)
        cur = torch.cuda.current_stream(dev)
        s_copy = _copy_stream(dev)
        # prenet dropout (on at inference, model.py:95-100): masks drawn on the GPU, seeded by the sentence's GLOBAL index.
        # The mel stays in HBM: [len(ok), Fmax, 80] on the device, rows past a sentence's frames zero.
        seeds = None if dropout_seed is None else [dropout_seed + mine[k] for k in ok]
        enc_kw = {}
        if enc_all is not None:
            enc_kw["encoded"] = enc_all.index_select(0, torch.tensor(ok, dtype=torch.long, device=enc_all.device))  # a device-side gather (plumbing)
        mel_dev = acoustic_model([toks[k] for k in ok], [frames[k] for k in ok], [nfr[k] for k in ok], dropout_seeds=seeds, to_host=False, **enc_kw)
        t_last = mark("acoustic_s", t_last)
        # the generator takes ragged batches (vtts_hifigan_forward_ragged: each utterance's samples are those of running it
        # alone): the sentences sorted by length, cut into passes by _generator_batches, each cut to its longest
        pending = []
        pf = _pass_frames(generator)
        todo = sorted((r for r in range(len(ok)) if gfr[ok[r]] > 0), key=lambda r: gfr[ok[r]])
        for rows in _generator_batches(todo, [gfr[ok[r]] for r in todo], gen_batch if ragged else 1, pf):
            fr = [gfr[ok[r]] for r in rows]
            # the pass's slot length: the longest sentence, rounded up to a multiple of 4 frames (the fp32 engine's MFMA transposed convolution takes
            # lengths that are multiples of 4; an odd slot sent ups_0 of every utterance of the pass through the generic kernel).  Rows past a
            # sentence's own frames are masked by frames[] inside the kernels, so the pad (zeros: mel_dev's tail, or zeros appended here) is never read as data.
            Ts = -(-max(fr) // 4) * 4
            batch = mel_dev[torch.tensor(rows, device=dev), : min(Ts, mel_dev.shape[1])]  # a device-side gather (plumbing)
            if batch.shape[1] < Ts:
                batch = torch.nn.functional.pad(batch, (0, 0, 0, Ts - batch.shape[1]))
            batch = batch.contiguous()
            w = generator.forward_ragged(batch, fr) if ragged else generator(batch)
            # pinned, on a copy stream: the next pass computes while this one's samples leave
            done = torch.cuda.Event()
            done.record(cur)
            t_pin = time.perf_counter()
            host = torch.empty(w.shape, dtype=w.dtype, pin_memory=True)
            if timing is not None:
                timing["pinned_alloc_s"] = timing.get("pinned_alloc_s", 0.0) + time.perf_counter() - t_pin  # ~0 when the caching host allocator has a block
            if os.environ.get("VTTS_PIPE_COPY_ON_CUR"):  # diagnostic switch: the read-back on the generator's own stream
                host.copy_(w, non_blocking=True)
            else:
                s_copy.wait_event(done)
                with torch.cuda.stream(s_copy):
                    host.copy_(w, non_blocking=True)
                w.record_stream(s_copy)
            pending.append((rows, fr, host))
        cur.wait_stream(s_copy)
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
