"""ORACLE — TEST INFRASTRUCTURE ONLY.  Mints ``tests/golden/bench_parity_grade.npz``: known answers for the two side legs ``bench.py`` times at
north_star's tolerance (``pipeline_256.parity_grade``, ``longform_10min.parity_grade``), so that the bench line can carry an in-run max-abs
without importing the oracle (only ``bench.py``'s ``cpu_baseline`` leg may).  Everything here is CPU arithmetic on seeded synthetic inputs —
it does not need /root/reference and travels as a fixture.      python -m oracle.make_bench_golden

* **long-form** (BASELINE configs[4]): ``synthetic_mel(1, 37500, 99)`` — the 10-minute utterance of the bench — through
  ``hifigan_oracle.generator_forward`` (fp64; vietTTS/hifigan/model.py:109-125) on three windows: the utterance's first frames, frames straddling
  the chunk seam at 512 * 37, and its last frames.  An output sample depends on mel frames within +-12.71 frames, so the
  oracle run on ``[lo - 13, hi + 13)`` (cut at the utterance's true ends, where the generator's own zero padding applies) gives the samples of
  frames ``[lo, hi)`` of the whole utterance exactly.
* **pipeline** (BASELINE configs[3]): the three shortest of the 26 transcript lines (sentence i of the bench's 256 = line i mod 26, dropout seed
  7 + i) through the whole oracle chain — ``nat_oracle.duration_model`` (fp32, as text2mel.py:22-34) -> the reference's frame rules
  (text2mel.py:78-79, :90-102) -> ``nat_oracle.acoustic_inference`` (fp64) on the per-sentence threefry keep masks -> ``hifigan_oracle`` (fp64) —
  i.e. vietTTS/synthesizer.py:33-39 restated end to end; the integer frame counts travel too (BASELINE: bit-exact).
"""
from __future__ import annotations

import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HALO = 13
T10, CHUNK, WIN = 37500, 512, 16
SIL, SEED, NSENT = 0.05, 7, 3


def longform_windows():
    from oracle.hifigan_oracle import generator_forward
    from viettts_amd.hifigan.config import V1
    from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params

    params = synthetic_params(V1, 4321, "scaled")
    mel = synthetic_mel(1, T10, 99)
    seam = CHUNK * 37
    out = {}
    for name, lo in (("start", 0), ("seam", seam - WIN // 2), ("end", T10 - WIN)):
        hi = lo + WIN
        a, b = max(0, lo - HALO), min(T10, hi + HALO)
        y = generator_forward(params, mel[:, a:b], V1, np.float64)[0, :, 0]
        out[f"lf_{name}_lo"] = np.int64(lo)
        out[f"lf_{name}_wave"] = y[256 * (lo - a) : 256 * (hi - a)].astype(np.float64)
    return out


def pipeline_chain():
    from oracle import nat_oracle as no
    from oracle.hifigan_oracle import generator_forward
    from viettts_amd.hifigan.config import V1
    from viettts_amd.hifigan.synth import synthetic_params
    from viettts_amd.nat import text2mel as t2m
    from viettts_amd.nat.synth import synthetic_acoustic_checkpoint, synthetic_duration_checkpoint, transcript_sentences

    tdir = os.path.join(REPO, "tests", "golden", "text")
    sents = transcript_sentences(26, os.path.join(tdir, "transcript.txt"), os.path.join(tdir, "lexicon.txt"))
    Pd, Sd = synthetic_duration_checkpoint()
    Pa, Sa = synthetic_acoustic_checkpoint()
    params = synthetic_params(V1, 4321, "scaled")
    pick = sorted(range(26), key=lambda i: (len(sents[i]), i))[:NSENT]
    out = {"pipe_sentences": np.array(pick, np.int64)}
    for i in pick:
        tok = np.array(sents[i])
        d = no.duration_model(Pd, Sd, tok, dtype=np.float32)
        fr, nfr, trail = t2m.frame_plan([sents[i]], [d], SIL)
        masks = no.threefry_keep_masks(SEED + i, nfr[0], 256)
        mel = no.acoustic_inference(Pa, Sa, tok, fr[0], nfr[0], prenet_masks=lambda f, m=masks: (m[f, 0], m[f, 1]), dtype=np.float64)
        g = nfr[0] - trail[0]
        y = generator_forward(params, mel[None, :g].astype(np.float32), V1, np.float64)[0, :, 0]
        out[f"pipe_{i}_frames"] = np.array([nfr[0], trail[0]], np.int64)
        out[f"pipe_{i}_tokens"] = tok.astype(np.int64)
        out[f"pipe_{i}_wave"] = y.astype(np.float64)
        print(f"sentence {i}: {len(tok)} tokens, {nfr[0]} frames ({trail[0]} trailing silence), {y.shape[0]} samples", flush=True)
    return out


def main():
    out = {}
    out.update(longform_windows())
    out.update(pipeline_chain())
    out["meta"] = np.array(f"T10={T10} chunk={CHUNK} window={WIN} halo={HALO}; silence_duration={SIL} dropout_seed={SEED}+i; weights synthetic_params(V1, 4321, 'scaled'), "
                           f"synthetic_duration_checkpoint(), synthetic_acoustic_checkpoint(); fp64 oracle")
    path = os.path.join(REPO, "tests", "golden", "bench_parity_grade.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
