"""Round 5: the packed-f32 miscompute of round 4 (profiles/r04_a_pkfma_findings.md), bisected on the failing artefact itself.  The acoustic model on a
high-priority stream of its own, the bf16 generator on an UNRELATED mel beside it; the mel compared with the plain, nothing-beside call.  The library
under test comes from VTTS_HIFIGAN_LIB (variants built by tools/experiments/r05/pkfma_bisect.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from viettts_amd.hifigan.config import V1
from viettts_amd.hifigan.generator import Generator
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params
from viettts_amd.nat import text2mel as t2m
from viettts_amd.nat.acoustic import AcousticModel
from viettts_amd.nat.duration import DurationModel
from viettts_amd.nat.synth import synthetic_acoustic_checkpoint, synthetic_duration_checkpoint, transcript_sentences

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tdir = os.path.join(R, "tests", "golden", "text")
n = 12
beside = sys.argv[1] if len(sys.argv) > 1 else "bf16"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sents = transcript_sentences(n, os.path.join(tdir, "transcript.txt"), os.path.join(tdir, "lexicon.txt"))
dm = DurationModel(device="cuda:0"); dm.load_params(*synthetic_duration_checkpoint())
am = AcousticModel(device="cuda:0"); am.load_params(*synthetic_acoustic_checkpoint())
opts = beside.split(":")[1:]  # e.g. bf16:fuse=0:streams=1, x3:fuse=1 — generator options of the aggressor
beside = beside.split(":")[0]
agg_kind = 0
if beside == "agg":  # a single-instruction-class aggressor (tools/kbench/pk_aggressor.hip): agg:kind=2
    import ctypes as C
    agg_kind = int(next(o.split("=")[1] for o in opts if o.startswith("kind=")))
    opts = []
    from viettts_amd import _lib as _l
    _l.load()
    agg = C.CDLL(os.path.join(R, "tools", "kbench", "bin", "libpk_aggressor.so"))
    agg.pk_aggressor_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    sink = torch.zeros(4, device="cuda:0")
gen = Generator(V1, device="cuda:0", dtype={"bf16": "bf16", "f32": "f32", "x3": "bf16x3", "none": "bf16", "agg": "bf16"}[beside]); gen.load_params(synthetic_params(V1, 4321, "scaled"))
for o in opts:
    if "=" in o and o.split("=")[0] not in ("module", "mel"):
        gen.set_option(o.split("=")[0], int(o.split("=")[1]))
module = next((o.split("=")[1] for o in opts if o.startswith("module=")), None)  # run ONE convolution module / pair in a loop instead of the whole generator
secs = dm(sents)
frames, nfr, trail = t2m.frame_plan(sents, secs, 0.05)
ok = sorted(range(n), key=lambda k: (-nfr[k], k))
args = ([sents[k] for k in ok], [frames[k] for k in ok], [nfr[k] for k in ok])
seeds = [7 + k for k in ok]
plain = am(*args, dropout_seeds=seeds, to_host=False).clone()
torch.cuda.synchronize()
again = am(*args, dropout_seeds=seeds, to_host=False)
torch.cuda.synchronize()
assert torch.equal(again, plain), "the plain call is not reproducible by itself"
other = torch.from_numpy(synthetic_mel(8, 200, 5)).to("cuda:0")
gen(other); torch.cuda.synchronize()
cur = torch.cuda.current_stream()
s_ac = torch.cuda.Stream(priority=-1)
tot_rows, even, odd, wrong_reps = 0, 0, 0, 0
for rep in range(reps):
    s_ac.wait_stream(cur)
    with torch.cuda.stream(s_ac):
        m = am(*args, dropout_seeds=seeds, to_host=False)
    m.record_stream(cur)
    if beside != "none":
        torch.cuda._sleep(int(8.0 * 2.0e6))  # let the decoder get ahead, then the generator beside it
        if agg_kind:
            for _ in range(8):
                assert agg.pk_aggressor_launch(C.c_void_p(cur.cuda_stream), agg_kind, 512, 6000, C.c_void_p(sink.data_ptr())) == 0
        elif module is None:
            for _ in range(6):
                gen(other)
        else:
            from viettts_amd.hifigan.weights import conv_specs
            spec = {sp.key: sp for sp in conv_specs(V1)}["generator/~/" + module.replace("+", "/~/")]
            L = 200 * {512: 1, 256: 8, 128: 64, 64: 128, 32: 256}[spec.cin if spec.kind == "conv" and spec.cin != 80 else 512]
            xin = torch.randn(8, L, spec.cin, device="cuda:0") if beside == "bf16" else torch.randn(8, spec.cin, L, device="cuda:0")
            for _ in range(60):
                gen.run_pair("generator/~/" + module.replace("+", "/~/"), xin) if "convs1" in module else gen.run_module("generator/~/" + module.replace("+", "/~/"), xin, 0.1)
    cur.wait_stream(s_ac)
    torch.cuda.synchronize()
    bad = []
    for i in range(n):
        d = (m[i] != plain[i]).any(dim=1).nonzero()
        if d.numel():
            bad.append((i, int(d.min())))
            even += i % 2 == 0
            odd += i % 2 == 1
    tot_rows += len(bad)
    wrong_reps += bool(bad)
    print(f"  rep {rep}: rows whose mel differs from the plain call (row, first frame): {bad}")
print(f"RESULT lib={os.path.basename(os.environ.get('VTTS_HIFIGAN_LIB', 'product'))} beside={sys.argv[1] if len(sys.argv) > 1 else beside}: {wrong_reps} of {reps} trials wrong; wrong rows: {even} even, {odd} odd")
