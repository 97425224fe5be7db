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
