"""Host logic around the path: text front end, integer frame counts, WAV writer, CLI parser."""
import numpy as np
import pytest

from viettts_amd.nat import text2mel as t2m
from viettts_amd.nat.config import FLAGS, load_phonemes_set
from viettts_amd.synthesizer import build_parser, nat_normalize_text
from viettts_amd.wavio import float_to_pcm16, read_wav, write_wav


def test_phoneme_inventory():
    ph = load_phonemes_set()
    assert ph[:4] == ["sil", "sp", "spn", " "] and len(ph) == 4 + 89  # vietTTS/nat/config.py:24-39
    assert len(set(ph)) == len(ph)
    assert ph[4] == "a" and ph[-1] == "ỹ" and ph.index("đ") == 4 + 39


def test_normalize_text_matches_reference_rules():
    # vietTTS/synthesizer.py:21-31 — expected strings derived by hand from those six substitutions
    assert nat_normalize_text("Xin chào, thế giới.") == "xin chào sil thế giới sil"
    assert nat_normalize_text('  "A"\nB:  c!?') == "a sil b sil c sil"
    assert nat_normalize_text("a . , : b") == "a sil b"


def test_text2tokens(tmp_path):
    lex = tmp_path / "lexicon.txt"
    lex.write_text("xin\t x i n\nchào\t c h à o\n", encoding="utf-8")
    ph = load_phonemes_set()
    toks = t2m.text2tokens("xin sil chào zzq", lex)
    want = [0] + [ph.index(c) for c in "xin"] + [3] + [0] + [ph.index(c) for c in "chào"] + [3] + [ph.index("q"), 3] + [0]
    assert toks == want  # 'z' is not a phoneme: dropped letter-wise (text2mel.py:52-56)


def test_integer_frame_counts_are_fp32_truncations():
    d = np.array([[0.2, 0.0317, 0.1234567, 0.5]], dtype=np.float32)
    frames = t2m.durations_to_frames(d)
    assert frames.dtype == np.float32
    assert np.array_equal(frames, (d * np.float32(16000)) / np.float32(256))
    assert t2m.n_frames_from_durations(d) == int(np.float32(np.sum(frames, dtype=np.float32)))
    assert t2m.n_frames_from_durations(np.array([[1.0, 1.0]], np.float32)) == 125
    assert t2m.trailing_silence_frames(np.array([[0.1, 0.2]], np.float32)) == int(float(np.float32(0.2)) * 16000 / 256)
    r = t2m.apply_duration_rules([0, 5, 3, 0], np.array([[0.01, 0.1, 0.3, 0.02]], np.float32), 0.2)
    assert np.allclose(r, [[0.2, 0.1, 0.0, 0.2]])
    r = t2m.apply_duration_rules([0, 5, 3, 0], np.array([[0.01, 0.1, 0.3, 0.02]], np.float32), -1.0)
    assert np.allclose(r, [[0.01, 0.1, 0.0, 0.02]])


def test_text2mel_surface(tmp_path):
    lex = tmp_path / "lexicon.txt"
    lex.write_text("a\t a\n", encoding="utf-8")
    t2m.set_mel_provider(None)
    t2m.set_duration_model(None)
    with pytest.raises(FileNotFoundError):  # no checkpoint under FLAGS.ckpt_dir: the reference's behaviour (text2mel.py:27)
        t2m.text2mel("a", lex)
    t2m.set_mel_provider(lambda tokens, lf, sd: np.zeros((1, len(tokens), 80), np.float32))
    try:
        mel = t2m.text2mel("a a", lex, silence_duration=0.2)
        assert mel.shape == (1, 6, 80) and mel.dtype == np.float32
    finally:
        t2m.set_mel_provider(None)


def test_wav_writer_roundtrip(tmp_path):
    x = np.array([0.0, 0.5, -0.5, 1.0, -1.0, 1.7, 1e-5], dtype=np.float32)
    assert float_to_pcm16(x).tolist() == [0, 16384, -16384, 32767, -32767, 32767, 0]
    write_wav(tmp_path / "a.wav", x, 16000)
    sr, pcm = read_wav(tmp_path / "a.wav")
    assert sr == 16000 and pcm.tolist() == float_to_pcm16(x).tolist()
    assert (tmp_path / "a.wav").stat().st_size == 44 + 2 * len(x)


def test_cli_flags_match_reference():
    a = build_parser().parse_args([])
    assert str(a.output) == "clip.wav" and a.sample_rate == 16000 and a.silence_duration == -1 and a.lexicon_file is None
    a = build_parser().parse_args(["--text", "x", "--output", "o.wav", "--sample-rate", "22050", "--silence-duration", "0.2", "--lexicon-file", "l.txt"])
    assert a.text == "x" and a.sample_rate == 22050 and a.silence_duration == 0.2 and a.lexicon_file == "l.txt"


# ---- pinned to the reference's OWN functions (oracle/make_text_golden.py lifts them from /root/reference by AST) ----
import hashlib
import json
from pathlib import Path

GOLD = Path(__file__).parent / "golden"


def _text_golden():
    with open(GOLD / "text_golden.json", encoding="utf-8") as f:
        return json.load(f)


def test_config_constants_equal_reference():
    g = _text_golden()
    assert load_phonemes_set() == g["phonemes"]
    assert FLAGS.special_phonemes == g["special_phonemes"]
    assert FLAGS.sil_index == g["sil_index"] and FLAGS.word_end_index == g["word_end_index"]
    for k, v in g["flags"].items():
        assert getattr(FLAGS, k) == v, k


def test_lexicon_equals_reference_loader():
    g = _text_golden()
    lex = t2m.load_lexicon(GOLD / "text" / "lexicon.txt")
    assert len(lex) == g["lexicon_entries"]
    dig = hashlib.sha256("\n".join(f"{k}\t{v}" for k, v in sorted(lex.items())).encode("utf-8")).hexdigest()
    assert dig == g["lexicon_sha256"]


def test_lexicon_malformed_line_raises_like_reference(tmp_path):
    lex = tmp_path / "lexicon.txt"
    lex.write_text("a\t a\nbroken line without a tab\n", encoding="utf-8")
    with pytest.raises(ValueError):  # the reference's dict(lines) (text2mel.py:19)
        t2m.load_lexicon(lex)


def test_normalize_and_tokens_bit_exact_vs_reference_functions():
    """Every transcript line of the reference's demo (scripts/quick_start.sh:11-12), the whole transcript as ONE --text
    argument, and adversarial strings: normalised text and token ids equal the reference's own functions' output; where
    the reference raises (a lexicon phoneme outside the set), so do we, with the same exception type."""
    g = _text_golden()
    lex = GOLD / "text" / "lexicon.txt"
    assert g["n_transcript_lines"] == 26 and len(g["cases"]) >= 39
    n_tok = 0
    for c in g["cases"]:
        norm = nat_normalize_text(c["raw"])
        assert norm == c["normalized"], c["raw"]
        if "error" in c:
            with pytest.raises(ValueError):
                t2m.text2tokens(norm, lex)
            assert c["error"] == "ValueError"
        else:
            assert t2m.text2tokens(norm, lex) == c["tokens"], c["raw"]
            n_tok += len(c["tokens"])
    assert n_tok > 1900


def test_frame_plan_equals_the_per_sentence_rules_bit_for_bit():
    """viettts_amd.nat.text2mel.frame_plan (the pipeline's batched host step) against the per-sentence functions that mirror
    text2mel.py:78-79, :90-102 — float32 frame vectors bit-identical, integer counts equal, on ragged random sentences."""
    rng = np.random.default_rng(11)
    toks, secs = [], []
    for _ in range(300):
        n = int(rng.integers(1, 140))
        t = [0] + [int(v) for v in rng.choice([0, 3, 7, 19, 55, 90], size=n)] + ([0] if rng.random() < 0.7 else [9])
        toks.append(t)
        secs.append(np.abs(rng.normal(0.08, 0.05, size=len(t))).astype(np.float32))
    for sd in (-1.0, 0.05, 0.2):
        frames, nfr, trail = t2m.frame_plan(toks, secs, sd)
        for i, (t, s) in enumerate(zip(toks, secs)):
            d = t2m.apply_duration_rules(t, s[None, :], sd)
            assert np.array_equal(frames[i], t2m.durations_to_frames(d)[0]) and frames[i].dtype == np.float32
            assert nfr[i] == t2m.n_frames_from_durations(d)
            assert trail[i] == (t2m.trailing_silence_frames(d) if t[-1] == FLAGS.sil_index else 0)


def test_normalised_text_equals_the_reference_cli_print():
    """The line the reference's CLI printed when it was executed (tests/golden/synthesizer_golden.npz, oracle/make_synth_golden.py)."""
    import json

    import numpy as np

    from viettts_amd.synthesizer import nat_normalize_text

    g = np.load(Path(__file__).parent / "golden" / "synthesizer_golden.npz")
    lines = json.loads(str(g["stdout"]))
    assert lines[0] == "Normalized text input: " + nat_normalize_text(str(g["text"]))
    assert lines[1] == "writing output to file out.wav"


@pytest.mark.skipif(not Path("/root/reference/vietTTS/synthesizer.py").exists(), reason="needs /root/reference (build container only)")
def test_reference_cli_reproduces_the_committed_fixture(tmp_path):
    """Run the reference's CLI again (oracle/make_synth_golden.py, over the haiku / jax stand-in) and compare with the committed fixture."""
    import os
    import subprocess
    import sys

    import numpy as np

    out = tmp_path / "again.npz"
    r = subprocess.run([sys.executable, str(Path(__file__).parents[1] / "oracle" / "make_synth_golden.py")], env=dict(os.environ, VTTS_SYNTH_GOLDEN_OUT=str(out)),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    a, b = np.load(Path(__file__).parent / "golden" / "synthesizer_golden.npz"), np.load(out)
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
