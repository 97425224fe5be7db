"""Diagnostic (round 4): where do the overlapped and the serial pipeline differ?  (1) acoustic model plain vs grouped hand-over, (2) the generator on
the SAME mel in one ragged batch vs per-group batches, (3) repeated runs of each."""
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
from viettts_amd.pipeline import _overlap_groups

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tdir = os.path.join(R, "tests", "golden", "text")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sents = transcript_sentences(n, os.path.join(tdir, "transcript.txt"), os.path.join(tdir, "lexicon.txt"))
dm = DurationModel(device="cuda:0"); dm.load_params(*synthetic_duration_checkpoint())
am = AcousticModel(device="cuda:0"); am.load_params(*synthetic_acoustic_checkpoint())
gen = Generator(V1, device="cuda:0", dtype="bf16"); gen.load_params(synthetic_params(V1, 4321, "scaled"))
secs = dm(sents)
frames, nfr, trail = t2m.frame_plan(sents, secs, 0.05)
ok = sorted(range(n), key=lambda k: (-nfr[k], k))
print("nfr (desc):", [nfr[k] for k in ok])
args = ([sents[k] for k in ok], [frames[k] for k in ok], [nfr[k] for k in ok])
seeds = [7 + k for k in ok]
plain = am(*args, dropout_seeds=seeds, to_host=False).clone()
torch.cuda.synchronize()
for ng in (1, 2, 3, 4):
    b = _overlap_groups([nfr[k] for k in ok], ng)
    for rep in range(3):
        m = am(*args, dropout_seeds=seeds, to_host=False, group_row0=b)
        torch.cuda.synchronize()
        bad = [(i, int((m[i] != plain[i]).sum())) for i in range(n) if not torch.equal(m[i], plain[i])]
        print(f"acoustic groups={ng} bounds={b} rep={rep}: rows differing from the plain call: {bad}")
p2 = am(*args, dropout_seeds=seeds, to_host=False)
torch.cuda.synchronize()
print("plain vs plain again:", bool(torch.equal(p2, plain)))
# generator: one ragged batch vs per-group batches on the same mel
gfr = [nfr[k] - trail[k] for k in ok]
order = sorted(range(n), key=lambda r: gfr[r])
def run(rows):
    fr = [gfr[r] for r in rows]
    batch = plain[torch.tensor(rows, device="cuda:0"), : max(fr)].contiguous()
    w = gen.forward_ragged(batch, fr)
    torch.cuda.synchronize()
    return {r: w[q, : 256 * fr[q]].clone() for q, r in enumerate(rows)}
whole = run(order)
for ng in (2, 3, 4):
    b = _overlap_groups([nfr[k] for k in ok], ng)
    bad = []
    for g in range(len(b) - 1):
        rows = sorted(range(b[g], b[g + 1]), key=lambda r: gfr[r])
        part = run(rows)
        bad += [(r, gfr[r], int((part[r] != whole[r]).sum()), float((part[r] - whole[r]).abs().max())) for r in rows if not torch.equal(part[r], whole[r])]
    print(f"generator groups={ng}: rows differing from the one-batch call: {bad}")
for r in order[:4]:
    alone = run([r])
    print("alone vs whole row", r, bool(torch.equal(alone[r], whole[r])))
