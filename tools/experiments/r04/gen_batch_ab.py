"""Experiment: the pipeline's generator stage as one ragged pass of 256 sentences against two / four passes (the read-back of a pass overlaps the next
pass's compute).  python tools/experiments/r04/gen_batch_ab.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

from viettts_amd.hifigan.config import V1
from viettts_amd.hifigan.generator import Generator
from viettts_amd.hifigan.synth import synthetic_params
from viettts_amd.nat.acoustic import AcousticModel
from viettts_amd.nat.duration import DurationModel
from viettts_amd.nat.synth import synthetic_acoustic_checkpoint, synthetic_duration_checkpoint, transcript_sentences
from viettts_amd.pipeline import synthesize_sentences

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
gen = Generator(V1, device="cuda:0", dtype="bf16")
gen.load_params(synthetic_params(V1, 4321, "scaled"))
dm = DurationModel(device="cuda:0")
dm.load_params(*synthetic_duration_checkpoint())
am = AcousticModel(device="cuda:0")
am.load_params(*synthetic_acoustic_checkpoint())
am.set_option("bf16x3", 1)
tdir = os.path.join(REPO, "tests", "golden", "text")
sents = transcript_sentences(256, os.path.join(tdir, "transcript.txt"), os.path.join(tdir, "lexicon.txt"))
for rep in range(2):
    for gb in (0, 128, 64):
        w = None
        for _ in range(3):
            del w
            tm = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            w = synthesize_sentences(sents, dm, am, gen, silence_duration=0.05, dropout_seed=7, timing=tm, gen_batch=gb)
            torch.cuda.synchronize()
            tot = time.perf_counter() - t0
        print(f"gen_batch {gb:3d}: total {tot * 1e3:.2f} ms, generator stage {tm['generator_s'] * 1e3:.2f} ms")
