"""Ragged batches on the engines that answer to the reference's 1e-4 (round 5): ``vtts_hifigan_forward_ragged`` on VTTS_F32 and VTTS_BF16X3
handles.  Every utterance of a batch of different lengths must get the zero padding it would see alone at EVERY layer
(vietTTS/hifigan/model.py:8-10 get_padding, lax "SAME" for the transposed convolutions) — the reference itself runs one utterance per call
(vietTTS/hifigan/mel2wave.py:20-41), so "the utterance alone" is the reference's semantics and the oracle is the judge."""
import numpy as np
import pytest
import torch

from oracle import hifigan_oracle as orc
from viettts_amd.hifigan.config import V1
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params

pytestmark = pytest.mark.gpu
BOUND = {"f32": 2e-5, "bf16x3": 5e-5}  # the engines' own asserted bounds (tests/test_gpu_parity.py, tests/test_gpu_x3.py); north_star: 1e-4


@pytest.fixture(scope="module")
def v1_params():
    return synthetic_params(V1, 4321, "scaled")


def _gen(dtype, params):
    from viettts_amd.hifigan.generator import Generator

    g = Generator(V1, device="cuda:0", dtype=dtype)
    g.load_params(params)
    return g


def _mel(frames, T, seed, junk):
    g = torch.Generator().manual_seed(seed)
    mel = torch.clamp(-5 + 2 * torch.randn(len(frames), T, 80, generator=g), -11.5129, 2.0)
    for b, n in enumerate(frames):
        mel[b, n:] = junk  # whatever sits past the end must not matter (NaN included: the masks select, they do not multiply)
    return mel


@pytest.mark.parametrize("dtype", ["f32", "bf16x3"])
@pytest.mark.parametrize("fuse", [2, 0, 3])
@pytest.mark.parametrize("T", [64, 47])
def test_ragged_rows_equal_each_utterance_alone(v1_params, dtype, fuse, T):
    """Lengths from one frame to several tiles, multiples of 4 and not; slot lengths that are a multiple of 4 (conv_pre / ups_0 on the MFMA
    kernels, float4 staging with per-column tail masks) and not (the generic ups_0 on the fp32 engine).  A row equals
      * the same utterance as a ONE-row ragged call in the same slot length, bit for bit (same kernels, other batch);
      * the plain entry point on the utterance alone (T = its own length): bit for bit on the split engine (no kernel of it depends on
        T % 4) and on the fp32 engine where both runs take the same first transposed convolution; ~1e-7 where they do not;
    the rest of the slot is zero."""
    gen = _gen(dtype, v1_params)
    gen.set_option("fuse", fuse)
    try:
        frames = [1, 2, 5, 13, 31, 32, 33, 47, 40, 3][: 10 if T >= 47 else 9]
        frames = [min(n, T) for n in frames]
        mel = _mel(frames, T, 99, float("nan")).to("cuda:0")
        got = gen.forward_ragged(mel, frames).cpu().numpy()
        assert np.isfinite(got).all()
        for b, n in enumerate(frames):
            row = gen.forward_ragged(mel[b : b + 1].contiguous(), [n]).cpu().numpy()[0]
            assert np.array_equal(got[b], row), (b, n)
            alone = gen(mel[b : b + 1, :n].contiguous()).cpu().numpy()[0]
            same_kernels = dtype == "bf16x3" and fuse >= 1 or (n % 4 == 0 and T % 4 == 0)
            if same_kernels:
                assert np.array_equal(got[b, : 256 * n], alone), (b, n)
            else:
                assert np.abs(got[b, : 256 * n] - alone).max() < 2e-6, (b, n)
            assert not got[b, 256 * n :].any()
    finally:
        gen.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16x3"])
def test_ragged_rows_against_the_oracle(v1_params, dtype, capsys):
    """The rows of one ragged batch against ``oracle.hifigan_oracle.generator_forward`` (fp64) on each utterance alone — the reference's
    one-utterance-per-call semantics — at the engine's parity bound."""
    gen = _gen(dtype, v1_params)
    try:
        frames = [1, 2, 3, 4, 5, 13, 29, 12]
        T = 32
        mel = _mel(frames, T, 5, 777.0)
        got = gen.forward_ragged(mel.to("cuda:0"), frames).cpu().numpy()
        worst = 0.0
        for b, n in enumerate(frames):
            want = orc.generator_forward(v1_params, mel[b : b + 1, :n].numpy(), V1, np.float64)[0, :, 0]
            worst = max(worst, float(np.abs(got[b, : 256 * n] - want).max()))
        with capsys.disabled():
            print(f"\n[{dtype} ragged rows vs the fp64 oracle, frames {frames}] max|dy| {worst:.3e}")
        assert worst < BOUND[dtype]
    finally:
        gen.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16x3"])
def test_ragged_across_micro_batches_streams_and_large_launches(v1_params, dtype):
    """More frames than the small-launch schedules take (sequential ResBlocks, XCD-aware tile order on the long stages), cut into micro-batches
    on one and two streams: every micro-batch must see ITS utterances' lengths."""
    gen = _gen(dtype, v1_params)
    try:
        rng = np.random.default_rng(5)
        frames = [int(v) for v in rng.integers(3, 97, size=37)]
        frames[7], frames[20] = 96, 1
        T = 96
        mel = _mel(frames, T, 7, -3.0).to("cuda:0")
        base = gen.forward_ragged(mel, frames).clone()
        for b in (0, 7, 16, 20, 36):
            row = gen.forward_ragged(mel[b : b + 1].contiguous(), [frames[b]])[0]
            assert torch.equal(base[b], row), b
            assert not bool(base[b, 256 * frames[b] :].any())
        for mb, streams in ((0, 1), (8, 1), (16, 2), (5, 3)):
            gen.set_option("microbatch", mb)
            gen.set_option("streams", streams)
            assert torch.equal(gen.forward_ragged(mel, frames), base), (mb, streams)
    finally:
        gen.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16x3"])
def test_ragged_long_utterances_take_the_xcd_tile_order(v1_params, dtype):
    """Utterances long enough that the pair kernels pad their grids to whole rounds of the 8 XCDs (>= 64 tiles per row): the per-utterance
    tile ranges must cover exactly each utterance's valid tiles."""
    gen = _gen(dtype, v1_params)
    try:
        frames = [700, 333, 1, 512]
        T = 700
        mel = _mel(frames, T, 11, 55.0).to("cuda:0")
        got = gen.forward_ragged(mel, frames)
        for b, n in enumerate(frames):
            if n % 4 == 0 or dtype == "bf16x3":
                assert torch.equal(got[b, : 256 * n], gen(mel[b : b + 1, :n].contiguous())[0]), (b, n)
            else:
                assert float((got[b, : 256 * n] - gen(mel[b : b + 1, :n].contiguous())[0]).abs().max()) < 2e-6, (b, n)
            assert not bool(got[b, 256 * n :].any())
    finally:
        gen.close()


def test_ragged_bad_counts_stay_inside_the_slot(v1_params):
    """Counts outside [0, T] are clamped by every kernel (include/vtts_hifigan.h): nothing outside the utterance's own slot is touched."""
    gen = _gen("bf16x3", v1_params)
    try:
        T = 16
        mel = _mel([16, 16, 16], T, 3, 0.0).to("cuda:0")
        fr = torch.tensor([-5, 1000, 7], dtype=torch.int32, device="cuda:0")
        got = gen.forward_ragged(mel, fr)
        full = gen(mel)
        assert not bool(got[0].any())
        assert torch.equal(got[1], full[1])
        assert torch.equal(got[2, : 256 * 7], gen.forward_ragged(mel[2:3].contiguous(), [7])[0, : 256 * 7])
    finally:
        gen.close()
