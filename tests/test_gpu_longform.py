"""Chunked long-form synthesis and the CLI on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from viettts_amd.hifigan.config import V1
from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gen():
    from viettts_amd.hifigan.generator import Generator

    g = Generator(V1, device="cuda:0")
    g.load_params(synthetic_params(V1, 4321, "scaled"))
    yield g
    g.close()


def test_chunked_equals_unchunked(gen):
    """BASELINE configs[4] acceptance: chunked == un-chunked on a shorter clip (here 1500 frames,
    chunks of 256 + 13-frame halos, ragged last chunk)."""
    from viettts_amd.longform import synthesize_chunked

    mel = torch.from_numpy(synthetic_mel(1, 1500, 3)).to("cuda:0")
    full = gen(mel)[0]
    timing = {}
    got = synthesize_chunked(gen, mel[0], chunk_frames=256, max_batch=4, timing=timing)
    assert got.shape == full.shape
    assert (got - full).abs().max().item() < 5e-6
    assert timing["chunks"] == 6 and 0 < timing["first_chunk_s"] <= timing["total_s"]
    # chunk-DP: two "ranks" computed separately cover the clip disjointly and add up to the same thing
    a = synthesize_chunked(gen, mel[0], chunk_frames=256, rank=0, world=2)
    b = synthesize_chunked(gen, mel[0], chunk_frames=256, rank=1, world=2)
    assert ((a != 0) & (b != 0)).sum().item() == 0
    assert (a + b - full).abs().max().item() < 5e-6


@pytest.fixture(scope="module")
def gen_bf16():
    from viettts_amd.hifigan.generator import Generator

    g = Generator(V1, device="cuda:0", dtype="bf16")
    g.load_params(synthetic_params(V1, 4321, "scaled"))
    yield g
    g.close()


def _snr_db(got, want):
    return float(10 * np.log10((want ** 2).mean() / max(((got - want) ** 2).mean(), 1e-300)))


@pytest.mark.parametrize("T,chunk,with_oracle", [(700, 128, True), (1500, 256, False)], ids=["T700c128-oracle", "T1500c256"])
def test_chunked_equals_unchunked_bf16_default_pass(gen_bf16, T, chunk, with_oracle, capsys):
    """What bench.py's ``longform_10min`` leg runs, at a length the oracle can follow: the **bf16** engine through ``synthesize_chunked``
    with its DEFAULTS (``max_batch = 0``: the first chunk alone, then every other chunk of one fed length in ONE full-size pass whose
    kept samples leave in one strided copy), a ragged last chunk, and the chunk-DP split over two ranks.  Against
    (a) the un-chunked bf16 output (a 13-frame halo covers the +-12.71-frame receptive field and a sample's arithmetic does not depend
        on where its tile lies, so the kept samples are expected bit-identical; asserted within the bf16 bound, reported),
    (b) ``oracle.hifigan_oracle.generator_forward`` in fp64 on the whole utterance — the reference's mel2wave semantics
        (vietTTS/hifigan/mel2wave.py:37-40), bf16 bounds of tests/test_gpu_bf16.py: max-abs < 0.03, waveform SNR > 35 dB."""
    from oracle.hifigan_oracle import generator_forward
    from viettts_amd.longform import synthesize_chunked

    params = synthetic_params(V1, 4321, "scaled")
    mel_h = synthetic_mel(1, T, 3)
    mel = torch.from_numpy(mel_h).to("cuda:0")
    assert gen_bf16.get_option("microbatch") == 0 and gen_bf16.get_option("streams") == 0
    full = gen_bf16(mel)[0].clone()
    timing = {}
    got = synthesize_chunked(gen_bf16, mel[0], chunk_frames=chunk, timing=timing)  # defaults: halo 13, max_batch 0
    assert got.shape == full.shape and timing["chunks"] == -(-T // chunk) and T % chunk != 0
    d_self = float((got - full).abs().max())
    n_diff = int((got != full).sum())
    with capsys.disabled():
        print(f"\n[bf16 long-form T={T} chunk={chunk}: {timing['chunks']} chunks] chunked vs un-chunked bf16: max|d| {d_self:.3e} ({n_diff} samples differ)")
    assert d_self < 0.03
    if with_oracle:  # ~10 s of numpy per 256 frames: the shorter case only
        want = generator_forward(params, mel_h, V1, np.float64)[0, :, 0]
        g64 = got.double().cpu().numpy()
        e_or, snr = float(np.abs(g64 - want).max()), _snr_db(g64, want)
        e_full = float(np.abs(full.double().cpu().numpy() - want).max())
        with capsys.disabled():
            print(f"[bf16 long-form T={T}] chunked vs fp64 oracle: max|d| {e_or:.3e}, SNR {snr:.1f} dB (un-chunked vs oracle {e_full:.3e})")
        assert e_or < 0.03 and snr > 35.0, (e_or, snr)
    assert bool(torch.isfinite(got).all()) and float(got.abs().max()) <= 1.0
    # chunk c -> rank c mod 2: the two ranks' pieces are disjoint and add up to the single-rank result, bit for bit
    a = synthesize_chunked(gen_bf16, mel[0], chunk_frames=chunk, rank=0, world=2)
    b = synthesize_chunked(gen_bf16, mel[0], chunk_frames=chunk, rank=1, world=2)
    assert ((a != 0) & (b != 0)).sum().item() == 0
    assert torch.equal(a + b, got)
    # an explicit small pass size walks the same chunks 3 at a time: same samples
    assert torch.equal(synthesize_chunked(gen_bf16, mel[0], chunk_frames=chunk, max_batch=3), got)


@pytest.mark.parametrize("dtype,bound", [("bf16x3", 5e-5), ("f32", 2e-5)])
def test_chunked_equals_unchunked_parity_grade(dtype, bound, capsys):
    """BASELINE configs[4] at north_star's tolerance (round 5): ``synthesize_chunked`` with its DEFAULTS on the split-operand engine (and the fp32
    engine) — first chunk alone, one full-size pass, ragged last chunk, chunk-DP over two ranks — against
    (a) the un-chunked output of the same engine (the 13-frame halo covers the +-12.71-frame receptive field: bit-identical on the split engine,
        whose kernels do not depend on the chunk's length; the fp32 engine routes a length that is no multiple of 4 through the generic first
        transposed convolution: ~1e-7),
    (b) ``oracle.hifigan_oracle.generator_forward`` in fp64 on the whole utterance (vietTTS/hifigan/mel2wave.py:37-40) at the engine's bound."""
    from oracle.hifigan_oracle import generator_forward
    from viettts_amd.hifigan.generator import Generator
    from viettts_amd.longform import synthesize_chunked

    params = synthetic_params(V1, 4321, "scaled")
    g = Generator(V1, device="cuda:0", dtype=dtype)
    g.load_params(params)
    try:
        T, chunk = 700, 128
        mel_h = synthetic_mel(1, T, 3)
        mel = torch.from_numpy(mel_h).to("cuda:0")
        full = g(mel)[0].clone()
        timing = {}
        got = synthesize_chunked(g, mel[0], chunk_frames=chunk, timing=timing)
        assert got.shape == full.shape and timing["chunks"] == -(-T // chunk)
        d_self, n_diff = float((got - full).abs().max()), int((got != full).sum())
        want = generator_forward(params, mel_h, V1, np.float64)[0, :, 0]
        e_or = float(np.abs(got.double().cpu().numpy() - want).max())
        with capsys.disabled():
            print(f"\n[{dtype} long-form T={T} chunk={chunk}] chunked vs un-chunked: max|d| {d_self:.3e} ({n_diff} samples differ); chunked vs fp64 oracle: {e_or:.3e}")
        if dtype == "bf16x3":
            assert torch.equal(got, full)
        assert d_self < 2e-6 and e_or < bound
        a = synthesize_chunked(g, mel[0], chunk_frames=chunk, rank=0, world=2)
        b = synthesize_chunked(g, mel[0], chunk_frames=chunk, rank=1, world=2)
        assert ((a != 0) & (b != 0)).sum().item() == 0 and torch.equal(a + b, got)
        assert torch.equal(synthesize_chunked(g, mel[0], chunk_frames=chunk, max_batch=3), got)
    finally:
        g.close()


def test_dp_setup_single_rank(gen):
    from viettts_amd import dist as vdist
    from viettts_amd.hifigan.generator import Generator

    g2 = Generator(V1, device="cuda:0")
    # a second rank would receive rank 0's packed blob over RCCL; adopting it must give identical output
    g2.adopt_packed(gen.packed_blob().clone())
    mel = torch.from_numpy(synthetic_mel(2, 20, 9)).to("cuda:0")
    assert torch.equal(g2(mel), gen(mel))
    g2.close()
    g3 = Generator(V1, device="cuda:0")
    vdist.setup_generator_dp(g3, lambda: synthetic_params(V1, 4321, "scaled"), vdist.RankInfo(0, 1, 0))
    assert torch.equal(g3(mel), gen(mel))
    g3.close()


def test_cli_mel_file(tmp_path):
    from viettts_amd.hifigan.weights import save_haiku_pickle
    from viettts_amd.wavio import float_to_pcm16, read_wav
    from oracle.hifigan_oracle import mel2wave_oracle

    params = synthetic_params(V1, 4321, "scaled")
    (tmp_path / "assets/hifigan").mkdir(parents=True)
    (tmp_path / "assets/infore/hifigan").mkdir(parents=True)
    (tmp_path / "assets/hifigan/config.json").write_text(open(os.path.join(REPO, "assets/hifigan/config.json")).read())
    save_haiku_pickle(tmp_path / "assets/infore/hifigan/hk_hifi.pickle", params)
    mel = synthetic_mel(1, 12, 4)
    np.save(tmp_path / "m.npy", mel[0])
    env = dict(os.environ, PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, "-m", "viettts_amd.synthesizer", "--mel-file", "m.npy", "--output", "o.wav", "--sample-rate", "16000"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "writing output to file o.wav" in r.stdout
    sr, pcm = read_wav(tmp_path / "o.wav")
    want = float_to_pcm16(mel2wave_oracle(params, mel, V1))
    assert sr == 16000 and len(pcm) == 12 * 256
    assert np.abs(pcm.astype(np.int32) - want.astype(np.int32)).max() <= 1


def test_cli_text_to_wav_equals_the_reference_cli_executed(tmp_path):
    """``python -m vietTTS.synthesizer --text ... --output ... --lexicon-file ... --silence-duration ...`` of THIS repo (text ->
    tokens -> durations and mel on the GPU -> HiFi-GAN on the GPU -> PCM16 WAV) against what the REFERENCE'S CLI produced for the
    same arguments and the same files under its CWD (tests/golden/synthesizer_golden.npz: vietTTS/synthesizer.py and everything it
    imports, run from /root/reference by oracle/make_synth_golden.py over oracle/haiku_shim.py; float64): same two printed lines,
    same number of samples, every PCM16 sample within 1 LSB."""
    import json

    from oracle.make_synth_golden import write_assets
    from viettts_amd.wavio import float_to_pcm16, read_wav

    g = np.load(os.path.join(REPO, "tests", "golden", "synthesizer_golden.npz"))
    assert write_assets(tmp_path) == str(g["nat_params_sha256"])
    env = dict(os.environ, PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, "-m", "vietTTS.synthesizer", "--text", str(g["text"]), "--output", "out.wav", "--lexicon-file",
                        os.path.join(REPO, str(g["lexicon"])), "--silence-duration", str(float(g["silence_duration"]))],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert [l for l in r.stdout.splitlines() if l.strip()] == json.loads(str(g["stdout"]))
    sr, pcm = read_wav(tmp_path / "out.wav")
    want = float_to_pcm16(g["wave"])
    assert sr == int(g["samplerate"]) and pcm.shape == want.shape  # the integer frame count of the whole chain: bit-exact
    d = np.abs(pcm.astype(np.int32) - want.astype(np.int32))
    print(f"[CLI vs the reference's CLI: {pcm.shape[0]} samples] max |d PCM16| {int(d.max())} LSB, {float((d > 0).mean()) * 100:.2f} % of samples differ; |pcm| max {int(np.abs(want).max())}")
    assert d.max() <= 1
