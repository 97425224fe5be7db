"""Diagnostic 2 (round 4): synthesize_sentences overlapped / serial, repeated, each sentence against the sentence run ALONE."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from viettts_amd.hifigan.config import V1
from viettts_amd.hifigan.generator import Generator
from viettts_amd.hifigan.synth import synthetic_params
from viettts_amd.nat import text2mel as t2m
from viettts_amd.nat.acoustic import AcousticModel
from viettts_amd.nat.duration import DurationModel
from viettts_amd.nat.synth import synthetic_acoustic_checkpoint, synthetic_duration_checkpoint, transcript_sentences
from viettts_amd.pipeline import synthesize_sentences

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tdir = os.path.join(R, "tests", "golden", "text")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sents = transcript_sentences(n, os.path.join(tdir, "transcript.txt"), os.path.join(tdir, "lexicon.txt"))
dm = DurationModel(device="cuda:0"); dm.load_params(*synthetic_duration_checkpoint())
am = AcousticModel(device="cuda:0"); am.load_params(*synthetic_acoustic_checkpoint())
gen = Generator(V1, device="cuda:0", dtype="bf16"); gen.load_params(synthetic_params(V1, 4321, "scaled"))
alone = {}
for i in range(n):
    secs = dm([sents[i]])
    fr, nfr, trail = t2m.frame_plan([sents[i]], secs, 0.05)
    g = nfr[0] - trail[0]
    mel = am([sents[i]], [fr[0]], [nfr[0]], dropout_seeds=[7 + i], to_host=False)
    alone[i] = gen.forward_ragged(mel[:, :g].contiguous(), [g])[0].cpu().numpy()
    print(i, "nfr", nfr[0], "trail", trail[0])
def check(tag, **kw):
    for rep in range(3):
        out = synthesize_sentences(sents, dm, am, gen, silence_duration=0.05, dropout_seed=7, **kw)
        bad = [(i, int((out[i] != alone[i]).sum()), int(np.nonzero(out[i] != alone[i])[0].min()) if (out[i] != alone[i]).any() else -1, out[i].shape[0]) for i in range(n) if not np.array_equal(out[i], alone[i])]
        print(f"{tag} rep {rep}: differing from alone (index, #samples, first differing sample, length): {bad}")
check("serial", overlap_groups=1)
check("overlap3", overlap_groups=3)
gen.set_option("chains", 0)
check("overlap3 chains=0", overlap_groups=3)
check("serial chains=0", overlap_groups=1)
gen.set_option("chains", 1)
os.environ["VTTS_PIPE_COPY_ON_CUR"] = "1"
check("overlap3 copy-on-cur", overlap_groups=3)
check("serial copy-on-cur", overlap_groups=1)
