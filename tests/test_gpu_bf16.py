"""bf16 path (BASELINE configs[2]) on the GPU.  BASELINE.json fixes a tolerance only for fp32 (1e-4);
for bf16 the tests (a) check every kernel against the oracle evaluated on the SAME bf16-rounded
operands — so that only fp32-accumulation order and the final bf16 rounding differ — and (b) report
and bound the end-to-end error against the fp64 golden vectors of the reference."""
import json

import numpy as np
import pytest
import torch

from oracle import hifigan_oracle as orc
from viettts_amd.hifigan.config import TINY, V1
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params
from viettts_amd.hifigan.weights import conv_specs

pytestmark = pytest.mark.gpu


def bf(x):
    """round-to-nearest-even to bf16, returned as float64"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


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

    g = Generator(V1, device=dev, dtype="bf16")
    g.load_params(v1_params)
    yield g
    g.close()


def _rel(got, ref):
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))


def _res_conv_cases():
    seen, out = set(), []
    for s in conv_specs(V1):
        if s.kind == "conv" and s.cin == s.cout:
            sig = (s.cin, s.k, s.dilation)
            if sig not in seen:
                seen.add(sig)
                out.append(s)
    return out


@pytest.mark.parametrize("spec", _res_conv_cases(), ids=lambda s: f"C{s.cin}k{s.k}d{s.dilation}")
def test_resblock_conv_kat_bf16(gen, v1_params, dev, spec):
    rng = np.random.default_rng(spec.cin * 1000 + spec.k * 10 + spec.dilation)
    B, L = 2, 700  # ragged: not a multiple of the 256/512 time tiles
    x = rng.standard_normal((B, L, spec.cin)).astype(np.float32) * 2.0
    res = rng.standard_normal((B, L, spec.cin)).astype(np.float32)
    w, b = v1_params[spec.key]["w"], v1_params[spec.key]["b"]
    xin = bf(orc.leaky_relu(bf(x), 0.1))  # kernel: bf16(lrelu(bf16 x)) while staging
    ref = orc.conv1d(xin, bf(w), b.astype(np.float64), spec.dilation, orc.get_padding(spec.k, spec.dilation)) + bf(res)
    y = gen.run_module(spec.key, torch.from_numpy(x).to(dev), 0.1, torch.from_numpy(res).to(dev)).cpu().numpy()
    # only the output's bf16 rounding (2^-9 relative) and fp32 accumulation order differ
    assert np.abs(y - ref).max() <= 2.0 ** -8 * np.abs(ref).max(), (np.abs(y - ref).max(), np.abs(ref).max())
    assert np.abs(y - bf(ref)).max() <= 2.0 ** -7 * np.abs(ref).max()


def _pair_cases():
    by = {s.key: s for s in conv_specs(V1)}
    return [(s, by[s.key.replace("convs1_", "convs2_")]) for s in _res_conv_cases() if "convs1_" in s.key]


@pytest.mark.parametrize("pair", _pair_cases(), ids=lambda p: f"C{p[0].cin}k{p[0].k}d{p[0].dilation}")
def test_fused_pair_kat_bf16(gen, v1_params, dev, pair):
    """x' = c2(lrelu(c1(lrelu(x)))) + x in ONE kernel; reference = oracle ops on the same bf16-rounded
    operands, with xt rounded to bf16 where the kernel rounds it (after bias + LeakyReLU)."""
    c1, c2 = pair
    rng = np.random.default_rng(c1.cin * 100 + c1.k * 10 + c1.dilation)
    B, L = 2, 1100  # several ragged tiles for every N1 (118..506 outputs per workgroup)
    x = rng.standard_normal((B, L, c1.cin)).astype(np.float32) * 2.0
    w1, b1 = v1_params[c1.key]["w"], v1_params[c1.key]["b"]
    w2, b2 = v1_params[c2.key]["w"], v1_params[c2.key]["b"]
    xin = bf(orc.leaky_relu(bf(x), 0.1))
    xt = orc.conv1d(xin, bf(w1), b1.astype(np.float64), c1.dilation, orc.get_padding(c1.k, c1.dilation))
    xt = bf(orc.leaky_relu(xt, 0.1))
    ref = orc.conv1d(xt, bf(w2), b2.astype(np.float64), 1, orc.get_padding(c2.k, 1)) + bf(x)
    y = gen.run_pair(c1.key, torch.from_numpy(x).to(dev)).cpu().numpy()
    err = np.abs(y - ref).max()
    # xt's bf16 rounding can flip by one ulp against the fp64 reference's (different fp32 summation
    # order), each flip moving the output by ~2^-9*|xt|*|w2|: allow 2^-7 of the output range
    assert err <= 2.0 ** -7 * np.abs(ref).max(), (err, np.abs(ref).max())


def test_fused_pair_rejects_non_pair_keys(gen, dev):
    from viettts_amd import _lib

    for key in ("generator/~/res_block1_0/~/convs2_0", "generator/~/ups_0", "generator/~/nope"):
        with pytest.raises(_lib.VttsError):
            gen.run_pair(key, torch.zeros((1, 64, 256), device=dev))


@pytest.mark.parametrize("L", [300, 1, 67], ids=lambda v: f"L{v}")
@pytest.mark.parametrize("fuse", [3, 0], ids=["register-streamed", "first-generation"])
@pytest.mark.parametrize("i", [0, 1, 2, 3])
def test_upsample_kat_bf16(gen, v1_params, dev, i, fuse, L):
    """The transposed convolutions (model.py:88-94) on both kernel generations (fuse = 3: kernels_bf16_up.hip for all four;
    fuse = 0: the 3-tap convolution of kernels_bf16.hip) against the zero-stuffed, un-flipped restatement on the same bf16
    operands; L = 1: one frame (both neighbours are padding), 67: ragged against every tile width, 300: several tiles, and a
    small launch (the register-streamed kernel then splits the output rows over several workgroups per tile)."""
    spec = [s for s in conv_specs(V1) if s.key == f"generator/~/ups_{i}"][0]
    rng = np.random.default_rng(40 + i)
    B = 2
    x = rng.standard_normal((B, L, spec.cin)).astype(np.float32) * 2.0
    w, b = v1_params[spec.key]["w"], v1_params[spec.key]["b"]
    ref = orc.conv1d_transpose(bf(x), bf(w), b.astype(np.float64), spec.stride)
    gen.set_option("fuse", fuse)
    try:
        y = gen.run_module(spec.key, torch.from_numpy(x).to(dev), 1.0).cpu().numpy()
    finally:
        gen.set_option("fuse", 2)
    assert y.shape == (B, L * spec.stride, spec.cout)
    assert np.abs(y - ref).max() <= 2.0 ** -8 * np.abs(ref).max()


@pytest.mark.parametrize("i", [2, 3])
def test_upsample_kat_bf16_large_launch_staged_epilogue(gen, v1_params, dev, i):
    """ups_2 / ups_3 on a launch of >= 512 tiles: the register-streamed kernel then keeps both row chunks in one workgroup and stores them
    through LDS as whole rows (kernels_bf16_up.hip: UTile::STAGE_OUT; the KAT above only reaches the split-chunk, direct-store path).
    Windows at the start, around tile seams and at the ragged end against the restatement on the same bf16 operands (model.py:88-94)."""
    spec = [s for s in conv_specs(V1) if s.key == f"generator/~/ups_{i}"][0]
    n1 = 256 if i == 2 else 512  # frames per tile (UT2 / UT3)
    B, L = 2, 300 * n1 + 77      # 602 tiles, a partial last tile
    rng = np.random.default_rng(90 + i)
    x = (rng.standard_normal((B, L, spec.cin)).astype(np.float32) * 2.0)
    w, b = v1_params[spec.key]["w"], v1_params[spec.key]["b"]
    y = gen.run_module(spec.key, torch.from_numpy(x).to(dev), 1.0).cpu().numpy()
    assert y.shape == (B, L * spec.stride, spec.cout)
    s_ = spec.stride
    for a, e in ((0, 700), (n1 - 40, n1 + 40), (150 * n1 - 300, 150 * n1 + 300), (L - 700, L)):
        lo, hi = max(a - 2, 0), min(e + 2, L)
        ref = orc.conv1d_transpose(bf(x[:, lo:hi]), bf(w), b.astype(np.float64), s_)
        # a frame reads its two neighbours: frames a .. e - 1 have theirs inside the slice (or, at the tensor's own ends, the same zero padding)
        got = y[:, a * s_ : e * s_]
        want = ref[:, (a - lo) * s_ : (e - lo) * s_]
        assert np.abs(got - want).max() <= 2.0 ** -8 * np.abs(ref).max(), (i, a, e)


def test_conv_pre_and_post_bf16(gen, v1_params, dev):
    mel = synthetic_mel(2, 300, 5)
    w, b = v1_params["generator/~/conv1_d"]["w"], v1_params["generator/~/conv1_d"]["b"]
    ref = orc.conv1d(bf(mel), bf(w), b.astype(np.float64), 1, 3)
    y = gen.run_module("generator/~/conv1_d", torch.from_numpy(mel).to(dev), 1.0).cpu().numpy()
    assert np.abs(y - ref).max() <= 2.0 ** -8 * np.abs(ref).max()
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 1000, 32)).astype(np.float32)
    w, b = v1_params["generator/~/conv1_d_1"]["w"], v1_params["generator/~/conv1_d_1"]["b"]
    ref = np.tanh(orc.conv1d(bf(x), w.astype(np.float64), b.astype(np.float64), 1, 3))[..., 0]  # fp32 weights
    y = gen.run_module("generator/~/conv1_d_1", torch.from_numpy(x).to(dev), 1.0).cpu().numpy()
    assert y.shape == (2, 1000)
    assert np.abs(y - ref).max() < 1e-5


@pytest.mark.parametrize("fuse", [3, 2, 1, 0], ids=["fused-resblocks-all", "fused-resblocks", "fused-pairs", "per-conv"])
def test_generator_bf16_vs_reference_golden(golden_dir, gen, dev, capsys, fuse):
    """End-to-end error of the bf16 path against the reference's fp64 output (reported, loosely bounded)."""
    meta = json.load(open(golden_dir / "golden_meta.json"))["cases"]
    worst = {}
    gen.set_option("fuse", fuse)
    for case in ("v1_scaled_T8", "v1_scaled_T37"):
        rec = meta[case]
        g = np.load(golden_dir / f"{case}.npz")
        mel = torch.from_numpy(synthetic_mel(rec["B"], rec["T"], rec["mseed"])).to(dev)
        wav, pre = gen.forward_tap(mel, "pre_tanh")
        torch.cuda.synchronize()
        e_y = float(np.abs(wav.cpu().numpy() - g["y64"]).max())
        e_p = float(np.abs(pre.cpu().numpy() - g["pre64"]).max())
        snr = 10 * np.log10((g["pre64"] ** 2).mean() / ((pre.cpu().numpy() - g["pre64"]) ** 2).mean())
        worst[case] = (e_y, e_p, float(snr))
        assert e_y < 0.05 and e_p < 0.05 and snr > 35.0, worst
    gen.set_option("fuse", 2)
    with capsys.disabled():
        print("\n[bf16 end-to-end vs fp64 reference] (max|dy|, max|dpre|, SNR dB):", worst)


def test_bf16_taps_vs_oracle(gen, v1_params, dev):
    mel = synthetic_mel(2, 9, 11)
    taps = []
    orc.generator_forward(v1_params, mel, V1, np.float64, taps=taps)
    taps = dict(taps)
    for name, slope in (("conv_pre", 0.1), ("ups_0", 1.0), ("mrf_0", 0.1), ("ups_3", 1.0), ("mrf_3", 0.01)):
        want = taps[name]  # NWC
        if slope != 1.0:
            want = orc.leaky_relu(want, slope)  # the bf16 path stores these already activated
        _, got = gen.forward_tap(torch.from_numpy(mel).to(dev), name)
        torch.cuda.synchronize()
        got = got.cpu().numpy().reshape(want.shape)
        assert _rel(got, want) < 0.05, (name, _rel(got, want))


@pytest.mark.parametrize("T", [1, 3, 37, 300], ids=lambda t: f"T{t}")
def test_fused_resblock_equals_pair_path(gen, dev, T):
    """The whole-ResBlock kernels (C = 32: k = 3, 7, 11; C = 64, 128: k = 3) against the pair-by-pair path (fuse = 1): same
    operands, same bf16 rounding points, only fp32 summation order differs -> a few output ulps.  T = 1, 3: utterances
    shorter than every halo; 37, 300: several ragged windows, both utterance edges.  Checked at every stage that has a
    fused ResBlock (taps mrf_1: C = 128, mrf_2: C = 64, mrf_3: C = 32) and at the waveform."""
    mel = torch.from_numpy(synthetic_mel(2, T, 31 + T)).to(dev)
    out = {}
    for fuse in (3, 2, 1):  # 3: the ResBlock kernel wherever it exists; 2: where it is the faster choice (default); 1: pairs only
        gen.set_option("fuse", fuse)
        out[fuse] = {}
        for tap in ("mrf_1", "mrf_2", "mrf_3"):
            wav, t = gen.forward_tap(mel, tap)
            torch.cuda.synchronize()
            out[fuse][tap] = t.cpu().numpy().copy()
        out[fuse]["wav"] = wav.cpu().numpy().copy()
    gen.set_option("fuse", 2)
    for fuse in (3, 2):
        for tap in ("mrf_1", "mrf_2", "mrf_3"):
            ref = out[1][tap]
            err = np.abs(out[fuse][tap] - ref).max()
            assert err <= 2.0 ** -6 * np.abs(ref).max(), (fuse, tap, err, np.abs(ref).max())
        assert np.abs(out[fuse]["wav"] - out[1]["wav"]).max() < 0.02


def test_bf16_batch_and_microbatch_invariance(gen, dev):
    mel = torch.from_numpy(synthetic_mel(5, 64, 21)).to(dev)
    base = gen(mel).clone()
    for mb in (1, 2, 5):
        gen.set_option("microbatch", mb)
        assert torch.equal(gen(mel), base)
    gen.set_option("microbatch", 0)
    assert torch.equal(gen(mel[3:4]), base[3:4])
    assert torch.isfinite(base).all() and base.abs().max().item() < 1.0


def test_bf16_rejects_unsupported_architecture(dev):
    from viettts_amd import _lib
    from viettts_amd.hifigan.generator import Generator

    with pytest.raises(_lib.VttsError):
        Generator(TINY, device=dev, dtype="bf16")


@pytest.mark.parametrize("fuse", [2, 1, 0])
def test_ragged_batch_equals_each_utterance_alone(fuse):
    """vtts_hifigan_forward_ragged: utterances of different lengths in one batch; every utterance's samples are bit for
    bit those of running it alone with T = its own length (per-utterance zero padding at every layer), the rest of its
    slot is zero.  Lengths straddle tile boundaries of several stages (1 frame ... several tiles)."""
    from viettts_amd.hifigan.generator import Generator

    gen = Generator(V1, device="cuda:0", dtype="bf16")
    gen.load_params(synthetic_params(V1, 4321, "scaled"))
    gen.set_option("fuse", fuse)
    try:
        frames = [1, 2, 5, 13, 31, 32, 33, 47, 64, 3]
        T = max(frames)
        g = torch.Generator().manual_seed(99)
        mel = torch.clamp(-5 + 2 * torch.randn(len(frames), T, 80, generator=g), -11.5129, 2.0)
        for b, n in enumerate(frames):
            mel[b, n:] = 123.0  # whatever sits past the end must not matter
        mel = mel.to("cuda:0")
        got = gen.forward_ragged(mel, frames).cpu().numpy()
        for b, n in enumerate(frames):
            alone = gen(mel[b : b + 1, :n].contiguous()).cpu().numpy()[0]
            assert np.array_equal(got[b, : 256 * n], alone), (b, n)
            assert not got[b, 256 * n :].any()
    finally:
        gen.close()


@pytest.mark.parametrize("tiles", [0, 1, 2])
def test_stage4_tail_fusion_is_bit_identical(tiles):
    """Option "tail" (default on, round 5): the generator's last pair launch (stage 4, k = 11) also runs conv_post + tanh (model.py:123-124) on its own
    rows — the stage output is never written, conv_post_bf16_k is not launched.  The values conv_post sees and the order of its fmaf chain are the
    un-fused path's, so the samples are BIT-IDENTICAL with the option off: plain batches (one frame ... several tiles), ragged batches, wide and narrow
    tiles, the parallel-ResBlock schedule of small launches and the sequential one; a tap forces the un-fused path and still matches."""
    from viettts_amd.hifigan.generator import Generator

    gen = Generator(V1, device="cuda:0", dtype="bf16")
    gen.load_params(synthetic_params(V1, 4321, "scaled"))
    gen.set_option("tiles", tiles)
    gen.set_option("stage", 0)  # (the whole-stage launch of round 6 has its own test below; this one is about the pair launch's tail)
    try:
        assert gen.get_option("tail") == 1
        for B, T in ((1, 1), (2, 5), (3, 37), (2, 300), (9, 260)):
            mel = torch.from_numpy(synthetic_mel(B, T, 70 + T)).to("cuda:0")
            frames = [max(1, T - 3 * b) for b in range(B)]
            gen.set_option("tail", 1)
            fused, fused_r = gen(mel).clone(), gen.forward_ragged(mel, frames).clone()
            tapped, pre = gen.forward_tap(mel, "pre_tanh")
            gen.set_option("tail", 0)
            plain, plain_r = gen(mel).clone(), gen.forward_ragged(mel, frames).clone()
            assert torch.equal(fused, plain), (B, T)
            assert torch.equal(fused_r, plain_r), (B, T)
            assert torch.equal(tapped, plain) and torch.equal(torch.tanh(pre), plain) or float((torch.tanh(pre) - plain).abs().max()) < 1e-6
            for b, n in enumerate(frames):
                assert not bool(fused_r[b, 256 * n :].any())
    finally:
        gen.close()


@pytest.mark.parametrize("tail", [1, 0])
def test_stage_kernel_is_bit_identical(tail):
    """Option "stage" (round 6; default OFF — it measured slower than the launches it replaces, profiles/r06_a_stage_kernel_findings.md): the generator's whole last stage — three ResBlock1 (k = 3, 7, 11) from one LDS-resident window of ups_3's
    output, the MRF sum in registers, mean, LeakyReLU(0.01), conv_post, tanh (model.py:112-124) — is ONE launch (kernels_bf16_stage.hip).  Same operations in
    the same order per element as the launch-per-ResBlock path (whole-ResBlock kernels at k = 3, 7, three pair launches at k = 11, conv_post inside the last
    or as its own kernel): the samples are BIT-IDENTICAL — one frame ... many windows (386 samples each: T = 2 already spans two), ragged batches whose
    utterances end inside a window, micro-batches on two streams, and a tap (which takes the other path) still matches."""
    from viettts_amd.hifigan.generator import Generator

    gen = Generator(V1, device="cuda:0", dtype="bf16")
    gen.load_params(synthetic_params(V1, 4321, "scaled"))
    gen.set_option("tail", tail)
    try:
        assert gen.get_option("stage") == 0
        for B, T in ((1, 1), (2, 2), (2, 5), (3, 37), (2, 300), (9, 260), (4, 1031)):
            mel = torch.from_numpy(synthetic_mel(B, T, 170 + T)).to("cuda:0")
            frames = [max(1, T - 3 * b - (T // 3) * (b % 2)) for b in range(B)]
            gen.set_option("stage", 1)
            fused, fused_r = gen(mel).clone(), gen.forward_ragged(mel, frames).clone()
            tapped, pre = gen.forward_tap(mel, "pre_tanh")
            gen.set_option("stage", 0)
            plain, plain_r = gen(mel).clone(), gen.forward_ragged(mel, frames).clone()
            assert torch.equal(fused, plain), (B, T, float((fused - plain).abs().max()))
            assert torch.equal(fused_r, plain_r), (B, T, float((fused_r - plain_r).abs().max()))
            assert torch.equal(tapped, plain)
            for b, n in enumerate(frames):
                assert not bool(fused_r[b, 256 * n :].any())
        # micro-batches on two streams: every micro-batch's windows see ITS utterances' lengths and its own scratch
        mel = torch.from_numpy(synthetic_mel(7, 90, 12)).to("cuda:0")
        frames = [90, 3, 77, 41, 90, 1, 64]
        gen.set_option("stage", 0)
        want = gen.forward_ragged(mel, frames).clone()
        gen.set_option("stage", 1)
        for mb, streams in ((0, 1), (2, 2), (3, 1)):
            gen.set_option("microbatch", mb)
            gen.set_option("streams", streams)
            assert torch.equal(gen.forward_ragged(mel, frames), want), (mb, streams)
    finally:
        gen.close()


def test_ragged_batch_across_micro_batches_and_streams():
    """The engine splits a large batch into micro-batches (optionally on several streams); every micro-batch must see ITS
    utterances' lengths."""
    from viettts_amd.hifigan.generator import Generator

    gen = Generator(V1, device="cuda:0", dtype="bf16")
    gen.load_params(synthetic_params(V1, 4321, "scaled"))
    try:
        rng = np.random.default_rng(5)
        frames = [int(v) for v in rng.integers(3, 97, size=37)]
        T = max(frames)
        g = torch.Generator().manual_seed(7)
        mel = torch.clamp(-5 + 2 * torch.randn(len(frames), T, 80, generator=g), -11.5129, 2.0).to("cuda:0")
        want = {b: gen(mel[b : b + 1, : frames[b]].contiguous()).cpu().numpy()[0] for b in (0, 9, 16, 17, 31, 36)}
        for mb, streams in ((0, 1), (8, 1), (16, 2)):
            gen.set_option("microbatch", mb)
            gen.set_option("streams", streams)
            got = gen.forward_ragged(mel, frames).cpu().numpy()
            for b, w in want.items():
                assert np.array_equal(got[b, : 256 * frames[b]], w), (mb, streams, b)
                assert not got[b, 256 * frames[b] :].any()
    finally:
        gen.close()


# ------------------------------------------------------------------------------------------------------------------
# Parity at the BENCHMARK's own shapes (BASELINE configs[1] length T=512, configs[2] B=64 x T=1024, bf16).
# BASELINE.json fixes a tolerance for fp32 only (1e-4); for bf16 the bounds below are this repo's own, stated here and
# reported in bench.py's JSON line next to the throughput (`parity_bf16`): max-abs <= 0.03 on the tanh output and
# SNR >= 38 dB on the pre-tanh signal against the reference generator's fp64 output (measured: ~1e-2, ~45 dB).
# ------------------------------------------------------------------------------------------------------------------
BF16_MAXABS, BF16_SNR_DB = 0.03, 38.0


def _bf16_err(wav, pre, g):
    idx = g["idx"]
    y, p = wav[:, idx].astype(np.float64), pre[:, idx].astype(np.float64)
    e_y, e_p = float(np.abs(y - g["y64"]).max()), float(np.abs(p - g["pre64"]).max())
    snr = float(10 * np.log10((g["pre64"] ** 2).mean() / ((p - g["pre64"]) ** 2).mean()))
    return e_y, e_p, snr


def test_bf16_T512_vs_reference_golden(golden_dir, gen, dev, capsys):
    """B=1 x T=512 (BASELINE configs[1]'s shape on the bf16 engine) against the reference's fp64 output."""
    rec = json.load(open(golden_dir / "golden_meta.json"))["cases"]["v1_scaled_T512"]
    g = np.load(golden_dir / "v1_scaled_T512.npz")
    mel = torch.from_numpy(synthetic_mel(1, 512, rec["mseed"])).to(dev)
    wav, pre = gen.forward_tap(mel, "pre_tanh")
    torch.cuda.synchronize()
    e_y, e_p, snr = _bf16_err(wav.cpu().numpy(), pre.cpu().numpy(), g)
    with capsys.disabled():
        print(f"\n[bf16 B=1 T=512 vs fp64 reference] max|dy| {e_y:.3e}  max|dpre| {e_p:.3e}  SNR {snr:.1f} dB")
    assert e_y < BF16_MAXABS and snr > BF16_SNR_DB, (e_y, e_p, snr)
    s = g["sum_y64"]
    y64 = wav.cpu().numpy().astype(np.float64)
    assert abs((y64 ** 2).sum() - s[2]) / s[2] < 2e-2  # whole-tensor energy, not only the strided sample


def test_bf16_benchmark_shape_B64_T1024(golden_dir, dev, capsys):
    """The headline configuration itself: bench.py's rank-0 batch (64 x 1024 frames, bf16, one pass).
    (a) rows 0, 37, 63 against the reference generator's fp64 output on those rows (golden minted by
        oracle/make_golden.py from the reference's torch generator);
    (b) a row of the batch is BIT-identical to the same utterance run alone, and to the ragged entry point at full length;
    (c) every sample finite and inside tanh's range."""
    from viettts_amd.hifigan.generator import Generator

    rec = json.load(open(golden_dir / "golden_meta.json"))["cases"]["v1_scaled_B64_T1024"]
    g = np.load(golden_dir / "v1_scaled_B64_T1024.npz")
    rows = rec["rows"]
    gen = Generator(V1, device=dev, dtype="bf16")
    gen.load_params(synthetic_params(V1, rec["wseed"], rec["kind"]))
    try:
        mel = torch.from_numpy(synthetic_mel(64, 1024, rec["mseed"])).to(dev)
        wav, pre = gen.forward_tap(mel, "pre_tanh")
        torch.cuda.synchronize()
        assert wav.shape == (64, 256 * 1024)
        assert bool(torch.isfinite(wav).all()) and float(wav.abs().max()) <= 1.0
        e_y, e_p, snr = _bf16_err(wav[rows].cpu().numpy(), pre[rows].cpu().numpy(), g)
        with capsys.disabled():
            print(f"\n[bf16 B=64 T=1024 rows {rows} vs fp64 reference] max|dy| {e_y:.3e}  max|dpre| {e_p:.3e}  SNR {snr:.1f} dB")
        assert e_y < BF16_MAXABS and snr > BF16_SNR_DB, (e_y, e_p, snr)
        plain = gen(mel)
        assert torch.equal(plain, wav)  # the tap entry point does not change the result
        for b in (0, 37, 63):
            alone = gen(mel[b : b + 1].contiguous())
            assert torch.equal(alone[0], plain[b]), b
        rag = gen.forward_ragged(mel[32:], [1024] * 32)
        assert torch.equal(rag, plain[32:])
    finally:
        gen.close()


def test_parallel_resblocks_schedule_is_bit_identical_bf16():
    """bf16 engine, small launches: the three ResBlocks of a stage on parallel streams with scratch of their own; only a ResBlock's
    last kernel touches the shared bf16 accumulator and those kernels are chained by events in the sequential order, so every sample
    sees the same additions and roundings as one-after-the-other.  Same bits: plain, ragged, every fuse level."""
    from viettts_amd.hifigan.generator import Generator

    dev = torch.device("cuda", 0)
    gen = Generator(V1, device=dev, dtype="bf16")
    gen.load_params(synthetic_params(V1, 4321, "scaled"))
    for fuse in (2, 1, 0, 3):
        gen.set_option("fuse", fuse)
        for B, T in ((1, 37), (2, 160), (1, 512), (3, 97)):
            mel = torch.from_numpy(synthetic_mel(B, T, 11)).to(dev)
            outs = {}
            for mode in (1, 0):
                gen.set_option("chains", mode)
                a = gen(mel).clone()
                a2 = gen(mel).clone()
                torch.cuda.synchronize()
                assert torch.equal(a, a2)
                outs[mode] = a
            assert torch.equal(outs[1], outs[0]), (fuse, B, T, float((outs[1] - outs[0]).abs().max()))
    gen.set_option("fuse", 2)
    mel = torch.from_numpy(synthetic_mel(3, 120, 12)).to(dev)
    frames = [120, 33, 77]
    gen.set_option("chains", 1)
    r1 = gen.forward_ragged(mel, frames).clone()
    gen.set_option("chains", 0)
    r0 = gen.forward_ragged(mel, frames).clone()
    torch.cuda.synchronize()
    assert torch.equal(r1, r0)
    gen.close()


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_small_launch_graph_replay_is_bit_identical(dtype):
    """Small launches replay a captured hipGraph from the third call with the same buffers on (engine.hip: forward_maybe_graphed): same
    bits as the eager path, a graph per (buffers, B, T), dropped when an option or the weight blob changes, and transparent inside a
    caller's own capture."""
    from viettts_amd.hifigan.generator import Generator

    dev = torch.device("cuda", 0)
    gen = Generator(V1, device=dev, dtype=dtype)
    gen.load_params(synthetic_params(V1, 4321, "scaled"))
    mel = torch.from_numpy(synthetic_mel(1, 512, 5)).to(dev)
    mel2 = torch.from_numpy(synthetic_mel(2, 96, 6)).to(dev)
    out, out2 = torch.empty((1, 256 * 512), device=dev), torch.empty((2, 256 * 96), device=dev)
    gen.set_option("graph", 0)
    ref, ref2 = gen(mel, out).clone(), gen(mel2, out2).clone()
    gen.set_option("graph", 1)
    assert gen.get_option("graphs_cached") == 0
    for i in range(20):
        out.zero_()
        gen(mel, out)
        torch.cuda.synchronize()
        assert torch.equal(out, ref), i
        if i % 2:
            out2.zero_()
            gen(mel2, out2)
            torch.cuda.synchronize()
            assert torch.equal(out2, ref2), i
        if i == 5:
            assert gen.get_option("graphs_cached") == 0  # a few calls do not pay for a capture
    assert gen.get_option("graphs_cached") == 2  # both keys kept coming back (20 and 10 calls): captured on their 8th
    for _ in range(2):
        out2.zero_()
        gen(mel2, out2)
    torch.cuda.synchronize()
    assert torch.equal(out2, ref2) and gen.get_option("graphs_cached") == 2
    # new data in the same buffers: the graph reads the buffers, not a snapshot
    mel.copy_(torch.from_numpy(synthetic_mel(1, 512, 77)))
    gen(mel, out)
    g_out = out.clone()
    gen.set_option("graph", 0)  # bumps the epoch: the cached graphs are stale
    assert torch.equal(gen(mel, out), g_out)
    gen.set_option("graph", 1)
    for _ in range(10):
        gen(mel, out)
    torch.cuda.synchronize()
    assert torch.equal(out, g_out) and gen.get_option("graphs_cached") >= 1
    n_before = gen.get_option("graphs_cached")
    # large launches never go through a graph
    big = torch.from_numpy(synthetic_mel(8, 1024, 3)).to(dev)
    ob = torch.empty((8, 256 * 1024), device=dev)
    for _ in range(10):
        gen(big, ob)
    assert gen.get_option("graphs_cached") == n_before
    gen.close()


# ------------------------------------------------------------------------------------------------------------------
# Edge lengths against the ORACLE (not against the engine itself): utterances shorter than every halo, where every tile
# is an edge tile and takes the clamped / masked staging path (kernels_bf16_rbg.hip stage_x, kernels_bf16_rbk.hip) and
# the reference's zero padding (vietTTS/hifigan/model.py:8-10 get_padding, :109-125 Generator.__call__) decides most
# samples.  Same bounds as the benchmark-shape tests above (BF16_MAXABS on the waveform, BF16_SNR_DB on the pre-tanh
# signal), every `fuse` setting, both fused-pair tile widths (option "tiles": 1 wide, 2 narrow).
# ------------------------------------------------------------------------------------------------------------------
EDGE_FRAMES = [1, 2, 3, 4, 5, 13]


@pytest.fixture(scope="module")
def edge_oracle(v1_params):
    """fp64 oracle output (waveform, pre-tanh) of one seeded mel per edge length."""
    out = {}
    for T in EDGE_FRAMES:
        mel = synthetic_mel(1, T, 900 + T)
        y, pre = orc.generator_forward(v1_params, mel, V1, np.float64, return_pre_tanh=True)
        out[T] = (mel, y[0, :, 0], pre[0, :, 0])
    return out


def _edge_check(wav, pre, want_y, want_pre, what):
    e_y = float(np.abs(wav.astype(np.float64) - want_y).max())
    snr = float(10 * np.log10((want_pre ** 2).mean() / ((pre.astype(np.float64) - want_pre) ** 2).mean()))
    assert np.isfinite(wav).all() and e_y < BF16_MAXABS and snr > BF16_SNR_DB, (what, e_y, snr)
    return e_y, snr


@pytest.mark.parametrize("tiles", [1, 2], ids=["wide-tiles", "narrow-tiles"])
@pytest.mark.parametrize("fuse", [3, 2, 1, 0], ids=["fused-resblocks-all", "fused-resblocks", "fused-pairs", "per-conv"])
def test_bf16_edge_lengths_vs_oracle(gen, dev, edge_oracle, capsys, fuse, tiles):
    gen.set_option("fuse", fuse)
    gen.set_option("tiles", tiles)
    worst = (0.0, 1e9)
    try:
        for T in EDGE_FRAMES:
            mel, want_y, want_pre = edge_oracle[T]
            wav, pre = gen.forward_tap(torch.from_numpy(mel).to(dev), "pre_tanh")
            torch.cuda.synchronize()
            e_y, snr = _edge_check(wav.cpu().numpy()[0], pre.cpu().numpy()[0], want_y, want_pre, ("alone", T, fuse, tiles))
            worst = (max(worst[0], e_y), min(worst[1], snr))
        # the same utterances as rows of ONE ragged batch (vtts_hifigan_forward_ragged): each row against the oracle's
        # output for that utterance alone, the rest of the slot zero
        Tmax = max(EDGE_FRAMES)
        batch = np.full((len(EDGE_FRAMES), Tmax, V1.num_mels), 77.0, np.float32)  # whatever sits past an utterance's end must not matter
        for b, T in enumerate(EDGE_FRAMES):
            batch[b, :T] = edge_oracle[T][0][0]
        got = gen.forward_ragged(torch.from_numpy(batch).to(dev), EDGE_FRAMES).cpu().numpy()
        for b, T in enumerate(EDGE_FRAMES):
            want_y = edge_oracle[T][1]
            e_y = float(np.abs(got[b, : 256 * T].astype(np.float64) - want_y).max())
            assert e_y < BF16_MAXABS, ("ragged row", T, fuse, tiles, e_y)
            assert not got[b, 256 * T :].any()
    finally:
        gen.set_option("fuse", 2)
        gen.set_option("tiles", 0)
    with capsys.disabled():
        print(f"\n[bf16 edge lengths vs fp64 oracle, fuse={fuse} tiles={tiles}] worst max|dy| {worst[0]:.2e}, worst SNR {worst[1]:.1f} dB")


def test_resblock2_generator_bf16_vs_oracle(dev, capsys):
    """ResBlock2 generators (vietTTS/hifigan/model.py:54-74, config "resblock": "2", model.py:84) of the V1 shapes on the bf16 engine: two
    residual convolutions per block (x = c(leaky_relu(x)) + x, rates (1, 3)) on the per-convolution kernel, the MRF mean and the next
    layer's LeakyReLU in the last epilogue.  Against the fp64 oracle with the bf16 bounds of the ResBlock1 tests; rows of a batch and
    rows of a ragged batch against the utterance alone (bit-identical)."""
    import dataclasses

    from viettts_amd.hifigan.generator import Generator

    cfg = dataclasses.replace(V1, resblock="2", resblock_dilation_sizes=((1, 3), (1, 3), (1, 3)))
    cfg.validate()
    params = synthetic_params(cfg, 4321, "scaled")
    assert "generator/~/res_block1_0/~/conv1_d_1" in params and "generator/~/res_block1_0/~/convs1_0" not in params
    g = Generator(cfg, device=dev, dtype="bf16")
    g.load_params(params)
    try:
        outs = {}
        for T in (3, 13):
            mel = synthetic_mel(1, T, 500 + T)
            want_y, want_pre = orc.generator_forward(params, mel, cfg, np.float64, return_pre_tanh=True)
            wav, pre = g.forward_tap(torch.from_numpy(mel).to(dev), "pre_tanh")
            torch.cuda.synchronize()
            e_y, snr = _edge_check(wav.cpu().numpy()[0], pre.cpu().numpy()[0], want_y[0, :, 0], want_pre[0, :, 0], ("resblock2", T))
            outs[T] = (mel, wav.clone())
            with capsys.disabled():
                print(f"\n[bf16 ResBlock2 generator vs fp64 oracle, T={T}] max|dy| {e_y:.2e}, pre-tanh SNR {snr:.1f} dB")
        g.set_option("fuse", 0)  # the upsamplers / conv_pre on the first-generation kernel too (another summation order: bounds, not bits)
        wav0, pre0 = g.forward_tap(torch.from_numpy(outs[13][0]).to(dev), "pre_tanh")
        g.set_option("fuse", 2)
        want_y, want_pre = orc.generator_forward(params, outs[13][0], cfg, np.float64, return_pre_tanh=True)
        _edge_check(wav0.cpu().numpy()[0], pre0.cpu().numpy()[0], want_y[0, :, 0], want_pre[0, :, 0], ("resblock2 fuse=0", 13))
        batch = np.full((2, 13, cfg.num_mels), 55.0, np.float32)
        batch[0] = outs[13][0][0]
        batch[1, :3] = outs[3][0][0]
        got = g.forward_ragged(torch.from_numpy(batch).to(dev), [13, 3])
        torch.cuda.synchronize()
        assert torch.equal(got[0], outs[13][1][0]) and torch.equal(got[1, : 256 * 3], outs[3][1][0]) and not got[1, 256 * 3 :].any()
        both = g(torch.from_numpy(np.concatenate([outs[13][0], outs[13][0]])).to(dev))
        assert torch.equal(both[0], outs[13][1][0]) and torch.equal(both[1], outs[13][1][0])
    finally:
        g.close()
