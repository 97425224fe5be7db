"""Experiment (round 4, not shipped): the NAT decoder as one resident kernel per run of frames (csrc/nat.hip: nat_dec_persist_k, compiled only with
-DVTTS_NAT_PERSIST) against the per-frame launches: the same mel, bit for bit — one sentence, a narrow batch, wide batches across two sentence
tiles with sentences that finish early, runs cut at frame 64 and (forward_groups) at the groups' last frames.
    python viettts_amd/csrc/build.py --define VTTS_NAT_PERSIST=1 --libname libvtts_persist.so
    VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/libvtts_persist.so python tools/experiments/r04/persist_check.py
Result on MI355X (gpurun_out/r04_run20): identical; 26.7 ms against 22.4 for the acoustic model of 256 sentences (profiles/r04_e_nat_decoder_findings.md)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
import torch

from viettts_amd.nat.acoustic import AcousticModel
from viettts_amd.nat.synth import synthetic_acoustic_checkpoint


def case(seed, L):
    rng = np.random.default_rng(seed)
    tok = list(rng.integers(0, 100, size=L))
    dur = np.abs(rng.normal(3.0, 1.5, size=L)).astype(np.float32)
    dur[rng.integers(0, L)] = 0.0
    return tok, dur, max(1, int(np.sum(dur, dtype=np.float32)))


m = AcousticModel(device="cuda:0")
m.load_params(*synthetic_acoustic_checkpoint())


def both(cases, seeds, **kw):
    out = []
    for flag in ("0", "1"):
        os.environ["VTTS_NAT_PERSIST"] = flag
        out.append(m([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases], dropout_seeds=seeds, **kw))
    return out


for n, base in ((1, 700), (5, 710), (41, 720), (70, 800)):
    cases = [case(base + i, 2 + (i * 7) % 29) for i in range(n)]
    per_frame, resident = both(cases, [4000 + base + i for i in range(n)])
    for i in range(n):
        assert np.array_equal(per_frame[i], resident[i]), (n, i)
    print(f"{n} sentences, up to {max(c[2] for c in cases)} frames: identical")
cases = sorted((case(900 + i, 3 + (i * 5) % 23) for i in range(48)), key=lambda c: -c[2])
per_frame, resident = both(cases, [5000 + i for i in range(48)], to_host=False, group_row0=[0, 10, 30, 48])
torch.cuda.synchronize()
assert torch.equal(per_frame, resident)
print("48 sentences in three groups: identical")
