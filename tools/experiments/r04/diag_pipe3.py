"""Diagnostic 3 (round 4): what corrupts the acoustic model's late frames when the generator runs beside it?  The acoustic model on a stream of
its own (high / normal priority; grouped / plain), beside (a) the bf16 generator on an UNRELATED mel, (b) a dummy torch workload, (c) nothing;
the mel compared with the plain, un-overlapped call."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from viettts_amd.hifigan.config import V1
from viettts_amd.hifigan.generator import Generator
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params
from viettts_amd.nat import text2mel as t2m
from viettts_amd.nat.acoustic import AcousticModel
from viettts_amd.nat.duration import DurationModel
from viettts_amd.nat.synth import synthetic_acoustic_checkpoint, synthetic_duration_checkpoint, transcript_sentences
from viettts_amd.pipeline import _overlap_groups

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tdir = os.path.join(R, "tests", "golden", "text")
n = 12
sents = transcript_sentences(n, os.path.join(tdir, "transcript.txt"), os.path.join(tdir, "lexicon.txt"))
dm = DurationModel(device="cuda:0"); dm.load_params(*synthetic_duration_checkpoint())
am = AcousticModel(device="cuda:0"); am.load_params(*synthetic_acoustic_checkpoint())
gen = Generator(V1, device="cuda:0", dtype="bf16"); gen.load_params(synthetic_params(V1, 4321, "scaled"))
secs = dm(sents)
frames, nfr, trail = t2m.frame_plan(sents, secs, 0.05)
ok = sorted(range(n), key=lambda k: (-nfr[k], k))
args = ([sents[k] for k in ok], [frames[k] for k in ok], [nfr[k] for k in ok])
seeds = [7 + k for k in ok]
plain = am(*args, dropout_seeds=seeds, to_host=False).clone()
torch.cuda.synchronize()
other = torch.from_numpy(synthetic_mel(8, 200, 5)).to("cuda:0")
gen(other); torch.cuda.synchronize()
big = torch.randn(4096, 4096, device="cuda:0")
cur = torch.cuda.current_stream()
def trial(tag, prio, groups, beside, delay_ms=8.0):
    s_ac = torch.cuda.Stream(priority=prio)
    b = _overlap_groups([nfr[k] for k in ok], groups) if groups else None
    for rep in range(3):
        s_ac.wait_stream(cur)
        with torch.cuda.stream(s_ac):
            m = am(*args, dropout_seeds=seeds, to_host=False, group_row0=b) if b else am(*args, dropout_seeds=seeds, to_host=False)
        m.record_stream(cur)
        if beside == "gen":
            torch.cuda._sleep(int(delay_ms * 2.0e6))  # let the decoder get ahead, then the generator beside it
            for _ in range(6):
                gen(other)
        elif beside == "gen_ragged":
            torch.cuda._sleep(int(delay_ms * 2.0e6))
            for _ in range(6):
                gen.forward_ragged(other, [200, 180, 160, 150, 140, 120, 100, 90])
        elif beside == "torch":
            torch.cuda._sleep(int(delay_ms * 2.0e6))
            for _ in range(40):
                big @ big
        cur.wait_stream(s_ac)
        torch.cuda.synchronize()
        bad = []
        for i in range(n):
            d = (m[i] != plain[i]).any(dim=1).nonzero()
            if d.numel():
                bad.append((i, nfr[ok[i]], int(d.min()), int(d.numel())))
        print(f"{tag} rep {rep}: rows whose mel differs from the plain call (row, nfr, first frame, #frames): {bad}")
gen32 = Generator(V1, device="cuda:0", dtype="f32"); gen32.load_params(synthetic_params(V1, 4321, "scaled"))
gen32(other); torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == "short":
    _g = gen
    trial("hi-prio PLAIN, bf16 generator beside", -1, 0, "gen")
    trial("hi-prio grouped, ragged bf16 generator beside", -1, 3, "gen_ragged")
    gen = gen32
    trial("hi-prio PLAIN, FP32 generator beside", -1, 0, "gen")
    trial("normal-prio PLAIN, FP32 generator beside", 0, 0, "gen")
    sys.exit(0)
trial("hi-prio grouped, nothing beside", -1, 3, None)
trial("hi-prio grouped, generator beside", -1, 3, "gen")
trial("hi-prio grouped, ragged generator beside", -1, 3, "gen_ragged")
trial("hi-prio grouped, torch matmuls beside", -1, 3, "torch")
trial("normal-prio grouped, generator beside", 0, 3, "gen")
trial("hi-prio PLAIN, generator beside", -1, 0, "gen")
trial("normal-prio PLAIN, generator beside", 0, 0, "gen")
