"""The split-operand engine (VTTS_BF16X3, "bf16x3"): the fp32 engine's layouts, schedule and entry points with the ResBlock convolutions on the bf16
matrix pipe, every product formed from three bf16 x bf16 terms of two-term operand splits (viettts_amd/csrc/kernels_x3.hip).  It answers to
BASELINE.json's fp32 bar (1e-4 max-abs against the reference generator); the asserts here are tighter (5e-5) and print what is observed.
CPU emulation of the same arithmetic: tools/experiments/r04/split_precision_emulation.py (1.6-1.9e-5)."""
import json

import numpy as np
import pytest
import torch

from oracle import hifigan_oracle as orc
from viettts_amd.hifigan.config import V1
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params
from viettts_amd.hifigan.weights import conv_specs

pytestmark = pytest.mark.gpu
TOL = 1e-4    # north_star bar
BOUND = 5e-5  # what the split arithmetic is expected to meet with margin


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def v1_params():
    return synthetic_params(V1, 4321, "scaled")


@pytest.fixture(scope="module")
def gen(dev, v1_params):
    from viettts_amd.hifigan.generator import Generator

    g = Generator(V1, device=dev, dtype="bf16x3")
    g.load_params(v1_params)
    yield g
    g.close()


def _nwc(x):
    return np.ascontiguousarray(np.transpose(x, (0, 2, 1)))


def _pair_cases():
    seen, out = set(), []
    specs = conv_specs(V1)
    for i, s in enumerate(specs):
        if s.kind == "conv" and s.cin == s.cout and "convs1_" in s.key:
            sig = (s.cin, s.k, s.dilation)
            if sig not in seen:
                seen.add(sig)
                out.append((s, specs[i + 1]))
    return out


@pytest.mark.parametrize("pair", _pair_cases(), ids=lambda p: f"C{p[0].cin}k{p[0].k}d{p[0].dilation}")
def test_x3_pair_kat(gen, v1_params, dev, pair, capsys):
    """Every (C, k, rate) pair of V1 against the oracle's ``c2(lrelu(c1(lrelu(x)))) + x`` (vietTTS/hifigan/model.py:45-50) in fp64: lengths that are
    no multiple of any tile, several tiles per utterance.  Error relative to the output's magnitude: ~2^-17 per operand."""
    c1, c2 = pair
    rng = np.random.default_rng(c1.cin * 1000 + c1.k * 10 + c1.dilation)
    B, L = 2, {256: 301, 128: 611, 64: 1203, 32: 1203}[c1.cin]
    x = rng.standard_normal((B, c1.cin, L)).astype(np.float32) * 2.0
    xn = _nwc(x).astype(np.float64)
    w1, b1 = v1_params[c1.key]["w"].astype(np.float64), v1_params[c1.key]["b"].astype(np.float64)
    w2, b2 = v1_params[c2.key]["w"].astype(np.float64), v1_params[c2.key]["b"].astype(np.float64)
    xt = orc.conv1d(orc.leaky_relu(xn, 0.1), w1, b1, c1.dilation, orc.get_padding(c1.k, c1.dilation))
    ref = orc.conv1d(orc.leaky_relu(xt, 0.1), w2, b2, 1, orc.get_padding(c2.k, 1)) + xn
    y = gen.run_pair(c1.key, torch.from_numpy(x).to(dev))
    torch.cuda.synchronize()
    err = float(np.abs(_nwc(y.cpu().numpy()) - ref).max())
    rel = err / float(np.abs(ref).max())
    with capsys.disabled():
        print(f"\n[bf16x3 pair C={c1.cin} k={c1.k} d={c1.dilation}] max|err| {err:.3e} ({rel:.2e} of max|ref| {np.abs(ref).max():.2f})")
    assert rel < 3e-5, (err, rel)


@pytest.mark.parametrize("i", [0, 1, 2, 3])
@pytest.mark.parametrize("L", [37, 300])
def test_x3_upsample_kat(gen, v1_params, dev, i, L, capsys):
    """The four transposed convolutions with split operands (convt_x3_k) against the oracle's ``conv1d_transpose(lrelu(x, 0.1))`` in fp64
    (vietTTS/hifigan/model.py:112-114; lax "SAME"): ragged lengths, several tiles and the utterance edges."""
    spec = [s for s in conv_specs(V1) if s.key == f"generator/~/ups_{i}"][0]
    rng = np.random.default_rng(40 + i)
    B = 2
    x = rng.standard_normal((B, spec.cin, L)).astype(np.float32) * 2.0
    w, b = v1_params[spec.key]["w"], v1_params[spec.key]["b"]
    ref = orc.conv1d_transpose(orc.leaky_relu(_nwc(x).astype(np.float64), 0.1), w.astype(np.float64), b.astype(np.float64), spec.stride)
    y = gen.run_module(spec.key, torch.from_numpy(x).to(dev), 0.1)
    torch.cuda.synchronize()
    assert y.shape == (B, spec.cout, L * spec.stride)
    err = float(np.abs(_nwc(y.cpu().numpy()) - ref).max())
    rel = err / float(np.abs(ref).max())
    with capsys.disabled():
        print(f"\n[bf16x3 ups_{i} L={L}] max|err| {err:.3e} ({rel:.2e} of max|ref| {np.abs(ref).max():.2f})")
    assert rel < 3e-5, (err, rel)


@pytest.mark.parametrize("L", [1, 5, 64, 67, 300])
def test_x3_conv_pre_kat(gen, v1_params, dev, L, capsys):
    """conv_pre with split operands (conv_pre_x3_k, round 5) against the oracle's ``conv1d`` in fp64 (vietTTS/hifigan/model.py:83,110: 80 -> 512,
    k = 7, pad 3, no activation in front): one frame, fewer frames than the halo, a whole tile, a ragged second tile, several tiles."""
    spec = [s for s in conv_specs(V1) if s.key == "generator/~/conv1_d"][0]
    x = synthetic_mel(2, L, 50 + L)
    w, b = v1_params[spec.key]["w"].astype(np.float64), v1_params[spec.key]["b"].astype(np.float64)
    ref = orc.conv1d(x.astype(np.float64), w, b, 1, 3)
    y = gen.run_module(spec.key, torch.from_numpy(x).to(dev), 1.0)
    torch.cuda.synchronize()
    assert y.shape == (2, 512, L)
    err = float(np.abs(_nwc(y.cpu().numpy()) - ref).max())
    rel = err / float(np.abs(ref).max())
    with capsys.disabled():
        print(f"\n[bf16x3 conv_pre L={L}] max|err| {err:.3e} ({rel:.2e} of max|ref| {np.abs(ref).max():.2f})")
    assert rel < 3e-5, (err, rel)


@pytest.mark.parametrize("case", ["v1_scaled_T8", "v1_scaled_T37", "v1_scaled_T512"])
def test_x3_generator_vs_reference_golden(golden_dir, gen, dev, case, capsys):
    """Whole generator against the reference generator's own fp64 output (tests/golden, minted by oracle/make_golden.py from
    vietTTS/hifigan/torch_model.py through the reference's converter): the 1e-4 of BASELINE.json, asserted at 5e-5."""
    rec = json.load(open(golden_dir / "golden_meta.json"))["cases"][case]
    g = np.load(golden_dir / f"{case}.npz")
    mel = torch.from_numpy(synthetic_mel(rec["B"], rec["T"], rec["mseed"])).to(dev)
    wav, pre = gen.forward_tap(mel, "pre_tanh")
    torch.cuda.synchronize()
    wav, pre = wav.cpu().numpy().astype(np.float64), pre.cpu().numpy().astype(np.float64)
    if "idx" in g.files:
        wav, pre = wav[:, g["idx"]], pre[:, g["idx"]]
    e_y, e_p = float(np.abs(wav - g["y64"]).max()), float(np.abs(pre - g["pre64"]).max())
    with capsys.disabled():
        print(f"\n[bf16x3 {case} vs the reference generator, fp64] max|dy| {e_y:.3e}  max|d pre-tanh| {e_p:.3e}")
    assert e_y < BOUND and e_y < TOL and e_p < BOUND, (e_y, e_p)


def test_x3_headline_shape_and_row_independence(golden_dir, gen, dev, capsys):
    """64 x 1024 frames (the throughput shape, default two-stream schedule): rows 0, 37, 63 against the reference's fp64 output; a row of the batch
    is bit-identical to the utterance alone; finite, inside tanh's range."""
    rec = json.load(open(golden_dir / "golden_meta.json"))["cases"]["v1_scaled_B64_T1024"]
    g = np.load(golden_dir / "v1_scaled_B64_T1024.npz")
    rows, idx = rec["rows"], g["idx"]
    mel = torch.from_numpy(synthetic_mel(64, 1024, rec["mseed"])).to(dev)
    wav = gen(mel)
    torch.cuda.synchronize()
    y = wav[rows].cpu().numpy()[:, idx].astype(np.float64)
    e_y = float(np.abs(y - g["y64"]).max())
    with capsys.disabled():
        print(f"\n[bf16x3 B=64 T=1024 rows {rows} vs the reference generator, fp64] max|dy| {e_y:.3e}")
    assert e_y < BOUND and e_y < TOL
    assert bool(torch.isfinite(wav).all()) and float(wav.abs().max()) < 1.0
    for b in (0, 37, 63):
        assert torch.equal(gen(mel[b : b + 1].contiguous())[0], wav[b]), b


def test_x3_edge_lengths_and_microbatches(gen, v1_params, dev):
    for T in (1, 2, 3, 5):
        mel = synthetic_mel(1, T, 100 + T)
        want = orc.generator_forward(v1_params, mel, V1, np.float64)[..., 0]
        got = gen(torch.from_numpy(mel).to(dev)).cpu().numpy()
        assert got.shape == (1, 256 * T) and np.abs(got - want).max() < BOUND
    mel = torch.from_numpy(synthetic_mel(5, 64, 21)).to(dev)
    base = gen(mel).clone()
    for mb in (1, 2, 5):
        gen.set_option("microbatch", mb)
        assert torch.equal(gen(mel), base), mb
    gen.set_option("microbatch", 0)
    # fuse = 0 turns the split kernels off: the handle then IS the fp32 engine
    gen.set_option("fuse", 0)
    f32 = gen(mel).clone()
    gen.set_option("fuse", 2)
    assert not torch.equal(f32, base) and float((f32 - base).abs().max()) < BOUND


@pytest.mark.parametrize("T", [3, 37, 300])
def test_x3_whole_resblock_equals_the_pair_path(gen, dev, T):
    """resblock_x3_k (kernels_x3_rb.hip, round 5): the three pairs of a ResBlock and the MRF bookkeeping in one launch, the running x in registers.
    Per output element it performs the pair kernel's operations in the pair kernel's order, so the whole generator is BIT-IDENTICAL with the
    kernel off (fuse = 1: pairs only), at its default policy (fuse = 2: C = 32, and C = 64 at k = 3) and wherever it exists (fuse = 3) — on plain
    and on ragged batches, at the stage taps and on the waveform.  T = 3: windows shorter than every margin; 37, 300: several windows, both edges."""
    mel = torch.from_numpy(synthetic_mel(2, T, 300 + T)).to(dev)
    frames = [T, max(1, T - 2)]
    outs = {}
    try:
        for fuse in (1, 2, 3):
            gen.set_option("fuse", fuse)
            w = gen(mel).clone()
            _, m2 = gen.forward_tap(mel, "mrf_2")
            _, m3 = gen.forward_tap(mel, "mrf_3")
            outs[fuse] = (w, m2.clone(), m3.clone(), gen.forward_ragged(mel, frames).clone())
    finally:
        gen.set_option("fuse", 2)
    for fuse in (2, 3):
        for got, want, what in zip(outs[fuse], outs[1], ("wav", "mrf_2", "mrf_3", "ragged wav")):
            assert torch.equal(got, want), (fuse, what, float((got - want).abs().max()))


def test_x3_other_architectures_run_on_the_fp32_kernels(dev):
    """Shapes the split kernels do not cover (the TINY fixtures' channel counts, ResBlock2 generators) run on the fp32 engine's kernels under the
    same handle type: bit-identical to an fp32 handle."""
    from viettts_amd.hifigan.config import TINY, TINY2
    from viettts_amd.hifigan.generator import Generator

    for cfg in (TINY, TINY2):
        params = synthetic_params(cfg, 4321, "scaled")
        mel = torch.from_numpy(synthetic_mel(2, 12, 1234)).to(dev)
        outs = []
        for dt in ("f32", "bf16x3"):
            g = Generator(cfg, device=dev, dtype=dt)
            g.load_params(params)
            outs.append(g(mel).clone())
            g.close()
        assert torch.equal(outs[0], outs[1])


def test_mel2wave_dropin_on_the_split_engine(tmp_path, monkeypatch, v1_params):
    """``VTTS_MEL2WAVE_DTYPE=bf16x3``: the reference's ``mel2wave(mel)`` surface (vietTTS/hifigan/mel2wave.py:20-41) on the split-operand engine —
    same files read, same return type, inside the reference's 1e-4."""
    import os

    from viettts_amd.hifigan import mel2wave as m2w
    from viettts_amd.hifigan.weights import save_haiku_pickle

    (tmp_path / "assets/hifigan").mkdir(parents=True)
    (tmp_path / "assets/infore/hifigan").mkdir(parents=True)
    repo_cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets/hifigan/config.json")
    (tmp_path / "assets/hifigan/config.json").write_text(open(repo_cfg).read())
    save_haiku_pickle(tmp_path / "assets/infore/hifigan/hk_hifi.pickle", v1_params)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("VTTS_MEL2WAVE_DTYPE", "bf16x3")
    m2w.reload()
    try:
        mel = synthetic_mel(1, 40, 2)
        wav = m2w.mel2wave(mel)
        assert isinstance(wav, np.ndarray) and wav.dtype == np.float32 and wav.shape == (10240,)
        assert m2w._generator().dtype_name == "bf16x3"
        want = orc.mel2wave_oracle(v1_params, mel, V1)
        assert np.abs(wav - want).max() < BOUND
    finally:
        m2w.reload()
