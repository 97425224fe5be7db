"""BASELINE.json configs[3] on ONE GPU, stage by stage: 256 synthetic sentences (token ids of sentence-like lengths,
synthetic checkpoints) through duration model -> frame rules -> acoustic model -> HiFi-GAN (bf16).  Prints one JSON line
with per-stage wall times (device-synchronised) — a development measurement; the judged line is bench.py's."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viettts_amd.hifigan.config import V1  # noqa: E402
from viettts_amd.hifigan.generator import Generator  # noqa: E402
from viettts_amd.hifigan.synth import synthetic_params  # noqa: E402
from viettts_amd.nat import text2mel as t2m  # noqa: E402
from viettts_amd.nat.acoustic import AcousticModel  # noqa: E402
from viettts_amd.nat.config import FLAGS  # noqa: E402
from viettts_amd.nat.duration import DurationModel  # noqa: E402
from viettts_amd.nat.synth import synthetic_acoustic_checkpoint, synthetic_duration_checkpoint  # noqa: E402


def sentences(n=256, seed=2024):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        words = int(rng.integers(6, 22))
        body = []
        for _ in range(words):
            body += list(rng.integers(4, 90, size=int(rng.integers(2, 6)))) + [FLAGS.word_end_index]
        out.append([FLAGS.sil_index] + body + [FLAGS.sil_index])
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    sync = torch.cuda.synchronize
    dm = DurationModel(device="cuda:0")
    dm.load_params(*synthetic_duration_checkpoint())
    am = AcousticModel(device="cuda:0")
    am.load_params(*synthetic_acoustic_checkpoint())
    gen = Generator(V1, device="cuda:0", dtype="bf16")
    gen.load_params(synthetic_params(V1, 4321, "scaled"))
    sents = sentences(n)
    res = {}
    for rep in range(2):  # first pass warms allocators / code objects
        sync(); t0 = time.perf_counter()
        secs = dm(sents)
        sync(); t1 = time.perf_counter()
        frames, nfr = [], []
        for t, d in zip(sents, secs):
            d = t2m.apply_duration_rules(t, d[None, :], 0.05)
            frames.append(t2m.durations_to_frames(d)[0])
            nfr.append(max(1, t2m.n_frames_from_durations(d)))
        t2 = time.perf_counter()
        mels = am(sents, frames, nfr, dropout_seeds=[7 + i for i in range(len(sents))])
        sync(); t3 = time.perf_counter()
        order = sorted(range(len(mels)), key=lambda k: mels[k].shape[0])
        nsamp, nbatches = 0, 0
        for i0 in range(0, len(order), 64):
            ks = order[i0 : i0 + 64]
            fr = [mels[k].shape[0] for k in ks]
            batch = np.zeros((len(ks), max(fr), 80), dtype=np.float32)
            for r, k in enumerate(ks):
                batch[r, : fr[r]] = mels[k]
            w = gen.forward_ragged(torch.from_numpy(batch).to("cuda:0"), fr)
            nsamp += 256 * sum(fr)
            nbatches += 1
        sync(); t4 = time.perf_counter()
        res = {"sentences": n, "tokens": int(sum(map(len, sents))), "frames": int(sum(nfr)), "frames_max": int(max(nfr)), "samples": int(nsamp),
               "generator_batches": nbatches, "duration_s": t1 - t0, "host_rules_s": t2 - t1, "acoustic_s": t3 - t2, "generator_s": t4 - t3,
               "total_s": t4 - t0, "samples_per_s": nsamp / (t4 - t0)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
