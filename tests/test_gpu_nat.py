"""NAT duration model on the GPU through the C ABI vs the numpy oracle on the same (synthetic, seeded) checkpoint.
Tolerance: fp32 both sides, different summation orders and libm -> 2e-6 absolute on durations of ~0.1 s; the INTEGER
quantities the pipeline derives from them (text2mel.py:78-79, :99-101) must be identical (BASELINE.json)."""
import numpy as np
import pytest
import torch

from oracle import nat_oracle as no
from viettts_amd.nat import text2mel as t2m
from viettts_amd.nat.config import FLAGS
from viettts_amd.nat.synth import synthetic_duration_checkpoint

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    from viettts_amd.nat.duration import DurationModel

    assert torch.cuda.is_available()
    m = DurationModel(device="cuda:0")
    P, S = synthetic_duration_checkpoint()
    m.load_params(P, S)
    yield m, P, S
    m.close()


def test_duration_matches_oracle_ragged_batch(model):
    m, P, S = model
    rng = np.random.default_rng(11)
    sents = [list(rng.integers(0, 100, size=L)) for L in (1, 2, 3, 9, 37, 64, 120, 255, 256)]
    got = m(sents)
    worst = 0.0
    for s, g in zip(sents, got):
        ref = no.duration_model(P, S, np.array(s), dtype=np.float64)
        assert g.shape == ref.shape and g.dtype == np.float32
        worst = max(worst, float(np.abs(g - ref).max()))
    assert worst < 2e-6, worst


def test_rows_are_independent_and_batching_is_invariant(model):
    m, P, S = model
    rng = np.random.default_rng(12)
    sents = [list(rng.integers(0, 100, size=L)) for L in (50, 17, 80)]
    together = m(sents)
    for s, g in zip(sents, together):
        alone = m([s])[0]
        assert np.array_equal(alone, g)  # bit-exact: padding / batch composition must not leak into a row


def test_integer_frame_counts_equal_oracle(model):
    """256 synthetic sentences: n_frames and the trailing-silence frame count derived from GPU durations equal those
    derived from the oracle's fp32 durations."""
    m, P, S = model
    rng = np.random.default_rng(13)
    sents = []
    for _ in range(256):
        n = int(rng.integers(3, 60))
        body = list(rng.integers(4, 90, size=n))
        for k in range(4, n, 5):
            body[k] = FLAGS.word_end_index
        sents.append([FLAGS.sil_index] + body + [FLAGS.sil_index])
    got = m(sents)
    mismatches = 0
    for s, g in zip(sents, got):
        ref = no.duration_model(P, S, np.array(s), dtype=np.float32)
        dg = t2m.apply_duration_rules(s, g[None, :], 0.1)
        dr = t2m.apply_duration_rules(s, ref[None, :], 0.1)
        if t2m.n_frames_from_durations(dg) != t2m.n_frames_from_durations(dr) or t2m.trailing_silence_frames(dg) != t2m.trailing_silence_frames(dr):
            mismatches += 1
    assert mismatches == 0


def test_predict_duration_surface(model):
    m, P, S = model
    t2m.set_duration_model(m)
    try:
        tokens = [0, 5, 6, 3, 7, 3, 0]
        d = t2m.predict_duration(tokens)
        assert d.shape == (1, len(tokens)) and d.dtype == np.float32
        assert np.abs(d[0] - no.duration_model(P, S, np.array(tokens), dtype=np.float64)).max() < 2e-6
    finally:
        t2m.set_duration_model(None)


# ------------------------------------------------ acoustic model ------------------------------------------------
@pytest.fixture(scope="module")
def acoustic():
    from viettts_amd.nat.acoustic import AcousticModel
    from viettts_amd.nat.synth import synthetic_acoustic_checkpoint

    m = AcousticModel(device="cuda:0")
    P, S = synthetic_acoustic_checkpoint()
    m.load_params(P, S)
    yield m, P, S
    m.close()


def _case(seed, L):
    rng = np.random.default_rng(seed)
    tok = list(rng.integers(0, 100, size=L))
    dur = np.abs(rng.normal(3.0, 1.5, size=L)).astype(np.float32)  # frames per token
    dur[rng.integers(0, L)] = 0.0  # a word-end token (text2mel.py:95-97)
    nf = max(1, int(np.sum(dur, dtype=np.float32)))
    return tok, dur, nf


def test_acoustic_matches_oracle_no_dropout(acoustic):
    m, P, S = acoustic
    cases = [_case(21, 1), _case(22, 7), _case(23, 30)]
    got = m([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases])
    for (tok, dur, nf), g in zip(cases, got):
        ref = no.acoustic_inference(P, S, np.array(tok), dur, nf, dtype=np.float64)
        assert g.shape == ref.shape == (nf, 80)
        # an autoregressive fp32 loop over up to ~100 frames: errors of single steps (~1e-6) compound mildly
        assert np.abs(g - ref).max() < 5e-4 * max(1.0, np.abs(ref).max()), np.abs(g - ref).max()


def test_acoustic_matches_oracle_with_explicit_dropout_masks(acoustic):
    from viettts_amd.nat.acoustic import bernoulli_keep_masks

    m, P, S = acoustic
    tok, dur, nf = _case(24, 20)
    keep = bernoulli_keep_masks(nf, seed=5)
    g = m([tok], [dur], [nf], keep_masks=[keep])[0]
    ref = no.acoustic_inference(P, S, np.array(tok), dur, nf, prenet_masks=lambda t: (keep[t, 0], keep[t, 1]), dtype=np.float64)
    assert np.abs(g - ref).max() < 5e-4 * max(1.0, np.abs(ref).max())
    # and the masks matter: without them the mel differs
    g0 = m([tok], [dur], [nf])[0]
    assert np.abs(g - g0).max() > 1e-3


def test_acoustic_rows_independent(acoustic):
    m, P, S = acoustic
    cases = [_case(31, 9), _case(32, 25)]
    both = m([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases])
    for c, g in zip(cases, both):
        assert np.array_equal(m([c[0]], [c[1]], [c[2]])[0], g)


def test_text2mel_to_waveform_pipeline(model, acoustic, tmp_path):
    """BASELINE configs[3] in miniature on one GPU: text -> tokens -> durations (GPU) -> frame rules -> mel (GPU) ->
    HiFi-GAN (GPU), all through the reference's call surface, synthetic checkpoints."""
    from viettts_amd.hifigan.config import V1
    from viettts_amd.hifigan.generator import Generator
    from viettts_amd.hifigan.synth import synthetic_params

    dm, _, _ = model
    am, _, _ = acoustic
    lex = tmp_path / "lexicon.txt"
    lex.write_text("xin\tx i n\nchào\tc h à o\n", encoding="utf-8")
    t2m.set_duration_model(dm)
    t2m.set_acoustic_model(am)
    try:
        text = "xin chào sp việt nam"
        tokens = t2m.text2tokens(text, lex)
        mel = t2m.text2mel(text, lex, silence_duration=0.2)
        d = t2m.apply_duration_rules(tokens, t2m.predict_duration(tokens), 0.2)
        want_frames = t2m.n_frames_from_durations(d) - t2m.trailing_silence_frames(d)
        assert mel.shape == (1, want_frames, 80) and mel.dtype == np.float32 and np.isfinite(mel).all()
        assert np.array_equal(mel, t2m.text2mel(text, lex, silence_duration=0.2))  # same dropout seed -> same mel
        gen = Generator(V1, device="cuda:0", dtype="bf16")
        gen.load_params(synthetic_params(V1, 4321, "scaled"))
        wav = gen(torch.from_numpy(mel).to("cuda:0")).cpu().numpy()
        gen.close()
        assert wav.shape == (1, 256 * want_frames) and np.isfinite(wav).all() and np.abs(wav).max() < 1.0
    finally:
        t2m.set_duration_model(None)
        t2m.set_acoustic_model(None)


def test_text2mel_equals_the_reference_code_executed(tmp_path, monkeypatch):
    """The product's text2mel — checkpoints read from ``assets/infore/nat/*.pickle`` under the CWD as the reference does, both
    networks on the GPU, the checkpoint's rng behind the prenet dropout — against the mel the REFERENCE'S OWN text2mel.py /
    model.py produced for the same text, lexicon, silence_duration and checkpoints (tests/golden/nat_text2mel_golden.npz,
    minted by oracle/make_nat_golden.py under oracle/haiku_shim.py; float64): the frame counts are equal, the mel within fp32
    accumulation error of an autoregressive loop a few hundred frames long."""
    from pathlib import Path

    from oracle.make_nat_golden import write_checkpoints

    g = np.load(Path(__file__).parent / "golden" / "nat_text2mel_golden.npz")
    assert write_checkpoints(tmp_path) == str(g["params_sha256"])
    lexicon = Path(__file__).parent / "golden" / "text" / "lexicon.txt"
    monkeypatch.chdir(tmp_path)
    t2m.set_duration_model(None)
    t2m.set_acoustic_model(None)
    try:
        for ci in range(int(g["n_cases"])):
            p = f"c{ci}_"
            text, sil = str(g[p + "text"]), float(g[p + "silence_duration"])
            tokens = t2m.text2tokens(text, lexicon)
            assert tokens == [int(t) for t in g[p + "tokens"]]
            dur = t2m.predict_duration(tokens)
            assert np.abs(dur.astype(np.float64) - g[p + "durations_s"]).max() < 5e-6
            mel = t2m.text2mel(text, lexicon, sil)
            want = g[p + "mel_full"][: int(g[p + "n_frames"]) - int(g[p + "trailing_frames"])]
            assert mel.shape == (1,) + want.shape, (mel.shape, want.shape)  # integer frame counts: bit-exact
            err = float(np.abs(mel[0].astype(np.float64) - want).max())
            print(f"[text2mel vs the reference's code, case {ci}: {want.shape[0]} frames] max|d mel| {err:.2e} (|mel| max {np.abs(want).max():.2f})")
            assert err < 5e-5  # observed 3e-6
            # ... and with the acoustic model's bf16x3 option (include/vtts_nat.h): the same frames, the mel inside the SAME bar (observed 1.6e-5)
            am = t2m._ACOUSTIC_MODEL
            am.set_option("bf16x3", 1)
            try:
                mel3 = t2m.text2mel(text, lexicon, sil)
            finally:
                am.set_option("bf16x3", 0)
            err3 = float(np.abs(mel3[0].astype(np.float64) - want).max())
            print(f"[... with the bf16x3 option] max|d mel| {err3:.2e}")
            assert mel3.shape == mel.shape and err3 < 5e-5
    finally:
        t2m.set_duration_model(None)
        t2m.set_acoustic_model(None)


def test_integer_frame_counts_equal_the_reference_code_executed(tmp_path, monkeypatch):
    """BASELINE: "bit-exact for the NAT duration model's integer frame counts".  Every line of the reference's demo transcript: tokens ->
    durations on the GPU (one ragged batch) -> the silence rules and the two integer conversions, against the counts the REFERENCE'S OWN
    text2tokens / predict_duration / text2mel arithmetic produced (oracle/make_nat_golden.py over the haiku / jax stand-in, float32 as the
    reference computes; its float64 run agrees on all 26 lines).  A line whose frame sum lies within 2e-3 of an integer is reported, not
    asserted: there the result hinges on the order of the float32 additions (XLA's is not known)."""
    from pathlib import Path

    from oracle.make_nat_golden import write_checkpoints
    from viettts_amd.nat.duration import DurationModel

    g = np.load(Path(__file__).parent / "golden" / "nat_text2mel_golden.npz")
    assert write_checkpoints(tmp_path) == str(g["params_sha256"])
    lexicon = Path(__file__).parent / "golden" / "text" / "lexicon.txt"
    lines = [l.strip() for l in open(Path(__file__).parent / "golden" / "text" / "transcript.txt", encoding="utf-8") if l.strip()]
    params, state = t2m.load_duration_checkpoint(tmp_path / "assets/infore/nat/duration_latest_ckpt.pickle")
    dm = DurationModel()
    dm.load_params(params, state)
    toks = [t2m.text2tokens(l, lexicon) for l in lines]
    assert [len(t) for t in toks] == [int(n) for n in g["all_n_tokens"]]
    secs = dm(toks)
    _, nfr, trail = t2m.frame_plan(toks, secs, float(g["all_silence_duration"]))
    frac = g["all_frac_f64"]
    near = [i for i in range(len(lines)) if min(frac[i], 1.0 - frac[i]) < 2e-3]
    bad = [i for i in range(len(lines)) if i not in near and (nfr[i] != int(g["all_n_frames_f32"][i]) or trail[i] != int(g["all_trailing_f32"][i]))]
    agree_near = [i for i in near if nfr[i] == int(g["all_n_frames_f32"][i])]
    print(f"[frame counts vs the reference's code: {len(lines)} lines, {int(sum(nfr))} frames] equal on {len(lines) - len(near) - len(bad)} asserted lines; "
          f"near-integer lines {near}: equal on {agree_near}")
    assert not bad, [(i, nfr[i], int(g["all_n_frames_f32"][i])) for i in bad]
    assert len(near) <= 2


def test_pipeline_two_generator_passes_equal_one(model, acoustic, monkeypatch):
    """A large job that fits one generator pass leaves in two (viettts_amd/pipeline.py::_generator_batches: the first pass's read-back runs
    under the second's compute).  40 sentences with the frame threshold lowered so that the rule fires: the same samples, bit for bit, as
    the single pass."""
    from viettts_amd import pipeline
    from viettts_amd.hifigan.config import V1
    from viettts_amd.hifigan.generator import Generator
    from viettts_amd.hifigan.synth import synthetic_params

    dm, _, _ = model
    am, _, _ = acoustic
    rng = np.random.default_rng(43)
    sents = [[FLAGS.sil_index] + list(rng.integers(4, 90, size=int(rng.integers(2, 12)))) + [FLAGS.sil_index] for _ in range(40)]
    gen = Generator(V1, device="cuda:0", dtype="bf16")
    gen.load_params(synthetic_params(V1, 4321, "scaled"))
    try:
        one = pipeline.synthesize_sentences(sents, dm, am, gen, silence_duration=0.05, gen_batch=1000)
        calls = []
        real = pipeline._generator_batches

        def spy(*a, **k):
            out = real(*a, **k)
            calls.append(len(out))
            return out

        monkeypatch.setattr(pipeline, "SPLIT_MIN_FRAMES", 0)
        monkeypatch.setattr(pipeline, "_generator_batches", spy)
        two = pipeline.synthesize_sentences(sents, dm, am, gen, silence_duration=0.05)
        assert calls == [2]
        assert sorted(one) == sorted(two) == list(range(40))
        for i in range(40):
            assert np.array_equal(one[i], two[i]), i
    finally:
        gen.close()


def test_pipeline_sharded_equals_unsharded(model, acoustic):
    """configs[3] on one GPU: 24 sentences through the batched pipeline; the union of two ranks' shards equals the
    single-rank result bit for bit (rows are independent at every stage; no exchange step)."""
    from viettts_amd.hifigan.config import V1
    from viettts_amd.hifigan.generator import Generator
    from viettts_amd.hifigan.synth import synthetic_params
    from viettts_amd.pipeline import synthesize_sentences

    dm, _, _ = model
    am, _, _ = acoustic
    rng = np.random.default_rng(41)
    sents = [[FLAGS.sil_index] + list(rng.integers(4, 90, size=int(rng.integers(2, 12)))) + [FLAGS.sil_index] for _ in range(24)]
    gen = Generator(V1, device="cuda:0", dtype="bf16")
    gen.load_params(synthetic_params(V1, 4321, "scaled"))
    try:
        whole = synthesize_sentences(sents, dm, am, gen, silence_duration=0.05)
        parts = {}
        for r in range(2):
            parts.update(synthesize_sentences(sents, dm, am, gen, silence_duration=0.05, rank=r, world=2))
        assert sorted(parts) == sorted(whole) == list(range(24))
        for i in range(24):
            assert whole[i].dtype == np.float32 and whole[i].shape[0] % 256 == 0
            assert np.array_equal(whole[i], parts[i]), i
        assert sum(w.shape[0] for w in whole.values()) > 0
    finally:
        gen.close()


@pytest.mark.parametrize("nat_bf16x3", [False, True], ids=["acoustic-fp32", "acoustic-bf16x3"])
def test_pipeline_waveforms_equal_each_sentence_alone_and_the_oracle(model, acoustic, capsys, nat_bf16x3):
    """What bench.py's ``pipeline_256`` leg runs (with the acoustic model's bf16x3 option, as the bench's headline pipeline number, and without),
    against something other than itself.  12 sentences of the reference's demo transcript
    (token ids pinned to the reference's text2tokens: tests/test_frontend_cpu.py) through ``synthesize_sentences`` — length-sorted
    rows, masks seeded by the global sentence index, the generator's ragged passes, the pinned read-back:
      (a) the generator's pass size (all sentences in one ragged pass, or five per pass) does not change a sample;
      (b) every waveform is bit-identical to the same sentence run ALONE: duration model -> frame rules -> acoustic model -> forward_ragged;
      (c) 8 of them are within the bf16 bounds (tests/test_gpu_bf16.py: max-abs < 0.03; waveform SNR > 35 dB) of the ORACLE chain
          nat_oracle.duration_model -> the reference's frame rules -> nat_oracle.acoustic_inference on the same threefry masks ->
          hifigan_oracle.generator_forward (fp64) — vietTTS/synthesizer.py:33-39 = text2mel, then mel2wave — with equal frame counts."""
    from pathlib import Path

    from oracle.hifigan_oracle import generator_forward
    from viettts_amd.hifigan.config import V1
    from viettts_amd.hifigan.generator import Generator
    from viettts_amd.hifigan.synth import synthetic_params
    from viettts_amd.nat.synth import transcript_sentences
    from viettts_amd.pipeline import synthesize_sentences

    dm, Pd, Sd = model
    am, Pa, Sa = acoustic
    tdir = Path(__file__).parent / "golden" / "text"
    sents = transcript_sentences(12, tdir / "transcript.txt", tdir / "lexicon.txt")
    sil, seed = 0.05, 7
    params = synthetic_params(V1, 4321, "scaled")
    gen = Generator(V1, device="cuda:0", dtype="bf16")
    gen.load_params(params)
    am.set_option("bf16x3", int(nat_bf16x3))
    try:
        serial = synthesize_sentences(sents, dm, am, gen, silence_duration=sil, dropout_seed=seed)
        again = synthesize_sentences(sents, dm, am, gen, silence_duration=sil, dropout_seed=seed, gen_batch=5)
        assert sorted(again) == sorted(serial) == list(range(12))
        for i in range(12):
            assert np.array_equal(again[i], serial[i]), i  # (a)
        frames_gpu = {}
        for i in range(12):  # (b)
            secs = dm([sents[i]])
            fr, nfr, trail = t2m.frame_plan([sents[i]], secs, sil)
            g = nfr[0] - trail[0]
            frames_gpu[i] = (nfr[0], trail[0])
            mel = am([sents[i]], [fr[0]], [nfr[0]], dropout_seeds=[seed + i], to_host=False)
            w = gen.forward_ragged(mel[:, :g].contiguous(), [g])[0].cpu().numpy()
            assert serial[i].shape == (256 * g,) and np.array_equal(w, serial[i]), i
            assert np.array_equal(gen(mel[:, :g].contiguous())[0].cpu().numpy(), serial[i]), i  # ... and to the plain entry point
        worst_e, worst_snr = 0.0, 1e9
        short = sorted(range(12), key=lambda i: serial[i].shape[0])[:8]
        for i in short:  # (c)
            tok = np.array(sents[i])
            d = no.duration_model(Pd, Sd, tok, dtype=np.float32)
            fr, nfr, trail = t2m.frame_plan([sents[i]], [d], sil)
            assert (nfr[0], trail[0]) == frames_gpu[i], i  # integer frame counts: equal (BASELINE.json)
            masks = no.threefry_keep_masks(seed + i, nfr[0], 256)
            mel = no.acoustic_inference(Pa, Sa, tok, fr[0], nfr[0], prenet_masks=lambda f, m=masks: (m[f, 0], m[f, 1]), dtype=np.float64)
            g = nfr[0] - trail[0]
            want = generator_forward(params, mel[None, :g].astype(np.float32), V1, np.float64)[0, :, 0]
            got = serial[i].astype(np.float64)
            e = float(np.abs(got - want).max())
            snr = float(10 * np.log10((want ** 2).mean() / ((got - want) ** 2).mean()))
            worst_e, worst_snr = max(worst_e, e), min(worst_snr, snr)
            assert e < 0.03 and snr > 35.0, (i, e, snr)
        with capsys.disabled():
            print(f"\n[pipeline ({'bf16x3' if nat_bf16x3 else 'fp32'} acoustic model) vs the oracle chain, 8 sentences, "
                  f"{sum(serial[i].shape[0] for i in short)} samples] worst max|dy| {worst_e:.3e}, worst SNR {worst_snr:.1f} dB")
    finally:
        am.set_option("bf16x3", 0)
        gen.close()


@pytest.mark.parametrize("dtype,bound", [("bf16x3", 5e-5), ("f32", 2e-5)])
def test_pipeline_parity_grade_vocoders_against_the_oracle_on_the_same_mel(model, acoustic, capsys, dtype, bound):
    """BASELINE configs[3] at north_star's tolerance (round 5): the sentence pipeline with the vocoder on the split-operand engine (and on the fp32
    engine) — ragged passes on those engines — and the acoustic model in its fp32 mode (the one pinned to the reference's code at 5e-5).
      (a) every waveform is bit-identical to the same sentence run ALONE (duration model -> frame rules -> acoustic model -> forward_ragged) on the
          split engine; within 2e-6 on the fp32 engine (whose first transposed convolution depends on the slot length mod 4);
      (b) north_star's "outputs match the Haiku generator on identical mel inputs within 1e-4": for the 8 shortest of 12 transcript sentences the
          pipeline's waveform against ``oracle.hifigan_oracle.generator_forward`` (fp64; vietTTS/hifigan/mel2wave.py:37-40) on the SAME mel the GPU
          acoustic model produced, asserted at the engine's own bound (5e-5 split / 2e-5 fp32);
      (c) reported, and bounded at 1e-3: the end-to-end difference against the whole ORACLE chain (nat_oracle.duration_model -> the reference's
          frame rules -> nat_oracle.acoustic_inference in fp64 on the same threefry masks -> hifigan_oracle), vietTTS/synthesizer.py:33-39 —
          here the fp32 acoustic model's own ~3e-6 of the mel's range passes through the vocoder too — with equal integer frame counts."""
    from pathlib import Path

    from oracle.hifigan_oracle import generator_forward
    from viettts_amd.hifigan.config import V1
    from viettts_amd.hifigan.generator import Generator
    from viettts_amd.hifigan.synth import synthetic_params
    from viettts_amd.nat.synth import transcript_sentences
    from viettts_amd.pipeline import synthesize_sentences

    dm, Pd, Sd = model
    am, Pa, Sa = acoustic
    tdir = Path(__file__).parent / "golden" / "text"
    sents = transcript_sentences(12, tdir / "transcript.txt", tdir / "lexicon.txt")
    sil, seed = 0.05, 7
    params = synthetic_params(V1, 4321, "scaled")
    gen = Generator(V1, device="cuda:0", dtype=dtype)
    gen.load_params(params)
    assert am.get_option("bf16x3") == 0
    try:
        tm = {}
        serial = synthesize_sentences(sents, dm, am, gen, silence_duration=sil, dropout_seed=seed, timing=tm)
        assert sorted(serial) == list(range(12))
        mels, plans = {}, {}
        for i in range(12):  # (a)
            secs = dm([sents[i]])
            fr, nfr, trail = t2m.frame_plan([sents[i]], secs, sil)
            g = nfr[0] - trail[0]
            plans[i] = (nfr[0], trail[0])
            mel = am([sents[i]], [fr[0]], [nfr[0]], dropout_seeds=[seed + i], to_host=False)
            mels[i] = mel[:, :g].contiguous()
            w = gen.forward_ragged(mels[i], [g])[0].cpu().numpy()
            assert serial[i].shape == (256 * g,), i
            if dtype == "bf16x3":
                assert np.array_equal(w, serial[i]), i
            else:  # the fp32 engine's first transposed convolution takes the generic kernel when the SLOT length is no multiple of 4 (another summation order)
                assert np.abs(w - serial[i]).max() < 2e-6, i
        short = sorted(range(12), key=lambda i: serial[i].shape[0])[:8]
        worst_same, worst_chain = 0.0, 0.0
        for i in short:
            mel_gpu = mels[i].cpu().numpy()
            want = generator_forward(params, mel_gpu, V1, np.float64)[0, :, 0]  # (b): identical mel inputs
            e = float(np.abs(serial[i].astype(np.float64) - want).max())
            worst_same = max(worst_same, e)
            assert e < bound, (i, e)
            tok = np.array(sents[i])  # (c): the whole chain
            d = no.duration_model(Pd, Sd, tok, dtype=np.float32)
            fr, nfr, trail = t2m.frame_plan([sents[i]], [d], sil)
            assert (nfr[0], trail[0]) == plans[i], i
            masks = no.threefry_keep_masks(seed + i, nfr[0], 256)
            mel_or = no.acoustic_inference(Pa, Sa, tok, fr[0], nfr[0], prenet_masks=lambda f, m=masks: (m[f, 0], m[f, 1]), dtype=np.float64)
            g = nfr[0] - trail[0]
            chain = generator_forward(params, mel_or[None, :g].astype(np.float32), V1, np.float64)[0, :, 0]
            worst_chain = max(worst_chain, float(np.abs(serial[i].astype(np.float64) - chain).max()))
        with capsys.disabled():
            print(f"\n[pipeline, {dtype} vocoder + fp32 acoustic model, 8 sentences, {sum(serial[i].shape[0] for i in short)} samples] "
                  f"vs the oracle generator on the SAME GPU mel: worst max|dy| {worst_same:.3e} (bound {bound:g}, north_star 1e-4); "
                  f"vs the whole oracle chain: {worst_chain:.3e}")
        assert worst_chain < 1e-3
    finally:
        gen.close()


def test_device_keep_masks_equal_the_threefry_restatement(acoustic):
    """Masks drawn by the library on the GPU == oracle/nat_oracle.py::threefry_keep_masks (pinned by Random123's
    known answers on the CPU side), and a sentence's masks do not depend on the batch it is in."""
    am, _, _ = acoustic
    seeds = [0, 1, 2**40 + 17, 2**62 + 5]
    got = am.device_keep_masks(seeds, 37).cpu().numpy()
    assert got.shape == (4, 37, 2, 256) and got.dtype == np.uint8
    for i, sd in enumerate(seeds):
        assert np.array_equal(got[i].astype(bool), no.threefry_keep_masks(sd, 37, 256)), sd
    alone = am.device_keep_masks([seeds[2]], 37).cpu().numpy()[0]
    assert np.array_equal(alone, got[2])


def test_device_haiku_masks_equal_the_reference_stream_restatement(acoustic):
    """vtts_nat_acoustic_keep_masks_haiku: the reference's own mask stream (checkpoint rng -> Haiku key chain -> jax.random
    bernoulli on the classic threefry layout) drawn on the GPU == oracle/nat_oracle.py::haiku_prenet_keep_masks bit for bit,
    the same for every sentence of the batch; and the acoustic model run with ``dropout_rng`` equals the oracle run on those masks."""
    am, P, S = acoustic
    for rng in ((0, 0), (123456789, 42), (0xFFFFFFFF, 0x80000001)):
        got = am.device_keep_masks_haiku(rng, 3, 41).cpu().numpy()
        want = no.haiku_prenet_keep_masks(np.array(rng, dtype=np.uint32), 41, 256)
        assert got.shape == (3, 41, 2, 256) and got.dtype == np.uint8
        for b in range(3):
            assert np.array_equal(got[b].astype(bool), want), (rng, b)
    # the other counter layout (jax_threefry_partitionable, JAX >= 0.5's default; an unpinned restatement: oracle/nat_oracle.py): device ==
    # restatement, and it IS another stream
    for rng in ((0, 0), (123456789, 42)):
        got = am.device_keep_masks_haiku(rng, 2, 29, partitionable=True).cpu().numpy()
        want = no.haiku_prenet_keep_masks(np.array(rng, dtype=np.uint32), 29, 256, partitionable=True)
        for b in range(2):
            assert np.array_equal(got[b].astype(bool), want), (rng, b)
        assert not np.array_equal(want, no.haiku_prenet_keep_masks(np.array(rng, dtype=np.uint32), 29, 256))
        assert abs(float(want.mean()) - 0.5) < 0.02
    rng = np.array([2024, 7], dtype=np.uint32)
    toks = [[0, 5, 9, 3, 14, 22, 3, 0], [0, 31, 3, 0]]
    frames = [np.array([3.0, 2.5, 4.0, 0.0, 3.5, 2.0, 0.0, 2.0], np.float32), np.array([2.0, 5.5, 0.0, 3.0], np.float32)]
    nfr = [int(np.sum(f, dtype=np.float32)) for f in frames]
    got = am(toks, frames, nfr, dropout_rng=rng)
    masks = no.haiku_prenet_keep_masks(rng, max(nfr), 256)
    for i in range(2):
        want = no.acoustic_inference(P, S, np.array(toks[i]), frames[i], nfr[i], prenet_masks=lambda t: (masks[t, 0], masks[t, 1]), dtype=np.float64)
        assert got[i].shape == want.shape
        assert np.abs(got[i] - want).max() <= 5e-4 * max(1.0, np.abs(want).max()), i


def test_acoustic_with_device_masks_matches_oracle(acoustic):
    """The product path: dropout seeds in, masks drawn on the GPU; the oracle gets the same masks from the restatement."""
    am, P, S = acoustic
    rng = np.random.default_rng(77)
    sents = [list(rng.integers(0, 100, size=L)) for L in (7, 19)]
    durs = [np.abs(rng.normal(2.0, 0.7, size=len(s))).astype(np.float32) + 0.3 for s in sents]
    nfr = [int(np.sum(d, dtype=np.float32)) for d in durs]
    seeds = [901, 902]
    got = am(sents, durs, nfr, dropout_seeds=seeds)
    for s, d, n, sd, g in zip(sents, durs, nfr, seeds, got):
        masks = no.threefry_keep_masks(sd, n, 256)
        ref = no.acoustic_inference(P, S, np.array(s), d, n, prenet_masks=lambda f, m=masks: (m[f, 0], m[f, 1]))
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(g - ref).max() / scale < 5e-4


def test_acoustic_wide_batch_rows_equal_rows_alone(acoustic):
    """More than 32 sentences take the two-tiles-per-wave decoder step; every row still equals the row run alone (and
    the oracle), finished sentences idle while longer ones keep decoding."""
    m, P, S = acoustic
    cases = [_case(500 + i, 2 + (i * 7) % 13) for i in range(41)]
    seeds = [3000 + i for i in range(41)]
    got = m([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases], dropout_seeds=seeds)
    for i in (0, 17, 33, 40):
        tok, dur, nf = cases[i]
        assert np.array_equal(m([tok], [dur], [nf], dropout_seeds=[seeds[i]])[0], got[i]), i
    tok, dur, nf = cases[40]
    masks = no.threefry_keep_masks(seeds[40], nf, 256)
    ref = no.acoustic_inference(P, S, np.array(tok), dur, nf, prenet_masks=lambda f: (masks[f, 0], masks[f, 1]))
    assert np.abs(got[40] - ref).max() < 5e-4 * max(1.0, np.abs(ref).max())


def test_acoustic_bf16x3_option(acoustic, capsys):
    """Option "bf16x3" (include/vtts_nat.h): the matrix products of the decoder's LSTM steps, of the gate GEMM and of the postnet as three
    bf16 x bf16 terms on the bf16 matrix pipe.  The mel stays within 2e-4 of its range of the fp32 mode's (observed ~1e-5) and within the fp32
    mode's own bound of the fp64 oracle; rows stay independent of their batch (narrow and wide batches, the grouped hand-over); the option
    reads back, and an unknown key is refused."""
    m, P, S = acoustic
    cases = [_case(31, 9), _case(32, 30), _case(33, 1), _case(34, 17)]
    seeds = [61, 62, 63, 64]
    args = ([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases])
    ref = m(*args, dropout_seeds=seeds)
    assert m.get_option("bf16x3") == 0
    m.set_option("bf16x3", 1)
    try:
        assert m.get_option("bf16x3") == 1
        got = m(*args, dropout_seeds=seeds)
        alone = m([cases[1][0]], [cases[1][1]], [cases[1][2]], dropout_seeds=[seeds[1]])[0]
    finally:
        m.set_option("bf16x3", 0)
    assert np.array_equal(alone, got[1])
    worst = 0.0
    for (tok, dur, nf), g, r, sd in zip(cases, got, ref, seeds):
        scale = max(1.0, float(np.abs(r).max()))
        worst = max(worst, float(np.abs(g - r).max()) / scale)
        masks = no.threefry_keep_masks(sd, nf, 256)
        orc = no.acoustic_inference(P, S, np.array(tok), dur, nf, prenet_masks=lambda f: (masks[f, 0], masks[f, 1]))
        assert np.abs(g - orc).max() < 5e-4 * max(1.0, np.abs(orc).max())
    # a wide batch (two 32-sentence tiles per wave), sentences that finish early, the grouped hand-over
    wide = sorted((_case(940 + i, 2 + (i * 7) % 23) for i in range(41)), key=lambda c: -c[2])
    wseeds = [7000 + i for i in range(41)]
    wargs = ([c[0] for c in wide], [c[1] for c in wide], [c[2] for c in wide])
    wref = m(*wargs, dropout_seeds=wseeds)
    m.set_option("bf16x3", 1)
    try:
        wgot = m(*wargs, dropout_seeds=wseeds)
        for i in (0, 20, 40):
            assert np.array_equal(m([wide[i][0]], [wide[i][1]], [wide[i][2]], dropout_seeds=[wseeds[i]])[0], wgot[i]), i
        grouped = m(*wargs, dropout_seeds=wseeds, to_host=False, group_row0=[0, 9, 25, 41])
        import torch

        torch.cuda.synchronize()
        for i in (0, 20, 40):
            assert np.array_equal(grouped[i, : wide[i][2]].cpu().numpy(), wgot[i]), i
    finally:
        m.set_option("bf16x3", 0)
    for g, r in zip(wgot, wref):
        worst = max(worst, float(np.abs(g - r).max()) / max(1.0, float(np.abs(r).max())))
    with capsys.disabled():
        print(f"[acoustic model, bf16x3 option vs fp32 mode] max|d mel| / range {worst:.2e}")
    assert 0.0 < worst < 2e-4
    from viettts_amd._lib import VttsError

    with pytest.raises(VttsError):
        m.set_option("no_such_option", 1)
    with pytest.raises(VttsError):
        m.set_option("bf16x3", 2)


def test_acoustic_from_a_precomputed_encoder_output(acoustic):
    """vtts_nat_acoustic_encode + forward_from_encoder (AcousticModel.encode / ``encoded=``): the token encoder run ahead for a batch, its rows
    re-ordered and a few dropped, then the rest of the model from them — the same mel, bit for bit, as the one-call forward of those sentences
    (plain and with the grouped hand-over; a wider Lmax than the subset's own)."""
    import torch

    m, P, S = acoustic
    cases = [_case(1300 + i, 2 + (i * 5) % 19) for i in range(37)]
    enc = m.encode([c[0] for c in cases])
    assert enc.shape == (37, max(len(c[0]) for c in cases), 512)
    order = sorted((i for i in range(37) if i % 5 != 3), key=lambda i: (-cases[i][2], i))
    sub = [cases[i] for i in order]
    seeds = [9000 + i for i in order]
    args = ([c[0] for c in sub], [c[1] for c in sub], [c[2] for c in sub])
    rows = enc.index_select(0, torch.tensor(order, device=enc.device))
    want = m(*args, dropout_seeds=seeds)
    got = m(*args, dropout_seeds=seeds, encoded=rows)
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    b = len(order)
    grouped = m(*args, dropout_seeds=seeds, encoded=rows, to_host=False, group_row0=[0, b // 3, b])
    torch.cuda.synchronize()
    for i in (0, b // 2, b - 1):
        assert np.array_equal(grouped[i, : sub[i][2]].cpu().numpy(), want[i]), i
    with pytest.raises(ValueError):
        m(*args, dropout_seeds=seeds, encoded=rows[:, :1])


def test_nat_models_run_from_an_adopted_blob(model, acoustic):
    """The data-parallel start-up (viettts_amd.dist.setup_model_dp): a rank that never saw the checkpoint binds the
    packed blob rank 0 broadcast and computes the same durations and mel bit for bit."""
    from viettts_amd.nat.acoustic import AcousticModel
    from viettts_amd.nat.duration import DurationModel

    dm, _, _ = model
    am, _, _ = acoustic
    dm2, am2 = DurationModel(device="cuda:0"), AcousticModel(device="cuda:0")
    try:
        assert dm2.packed_bytes == dm.packed_blob().numel() and am2.packed_bytes == am.packed_blob().numel()
        dm2.adopt_packed(dm.packed_blob().clone())
        am2.adopt_packed(am.packed_blob().clone())
        sents = [[3, 9, 27, 5], [8, 1, 4, 4, 60, 2, 11]]
        for a, b in zip(dm(sents), dm2(sents)):
            assert np.array_equal(a, b)
        tok, dur, nf = _case(61, 12)
        assert np.array_equal(am([tok], [dur], [nf], dropout_seeds=[5])[0], am2([tok], [dur], [nf], dropout_seeds=[5])[0])
        with pytest.raises(ValueError):
            am2.adopt_packed(torch.zeros(16, dtype=torch.uint8, device="cuda:0"))
    finally:
        dm2.close()
        am2.close()
