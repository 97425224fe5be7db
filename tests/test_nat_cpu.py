"""NAT duration model, CPU side: the oracle is self-consistent (fp32 vs fp64), the synthetic checkpoint carries exactly
the arrays the C ABI asks for, and include/vtts_nat.h's symbols are all exported (no compute without a GPU)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from oracle import nat_oracle as no
from viettts_amd import _lib
from viettts_amd.nat import text2mel as t2m
from viettts_amd.nat.config import FLAGS
from viettts_amd.nat.synth import synthetic_duration_checkpoint

REPO = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    from viettts_amd.csrc.build import build

    build()
    return _lib.load()


def test_nat_header_symbols_all_exported(lib):
    header = (REPO / "include" / "vtts_nat.h").read_text()
    declared = set(re.findall(r"\b(vtts_nat_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.NAT_EXPORTS), declared ^ set(_lib.NAT_EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_param_table_matches_synthetic_checkpoint(lib):
    from viettts_amd.nat.duration import _lookup

    h = C.c_void_p(0)
    cfg = _lib.NatDurationCfg(256, 256)
    _lib.check(lib, lib.vtts_nat_duration_create(C.byref(cfg), 0, C.byref(h)))
    n = C.c_int(0)
    _lib.check(lib, lib.vtts_nat_duration_num_params(h, C.byref(n)))
    P, S = synthetic_duration_checkpoint()
    seen = 0
    for i in range(n.value):
        mod, name = C.c_char_p(), C.c_char_p()
        shape = (C.c_int64 * 3)()
        nd = C.c_int(0)
        _lib.check(lib, lib.vtts_nat_duration_param_info(h, i, C.byref(mod), C.byref(name), shape, C.byref(nd)))
        a = _lookup(S if name.value == b"average" else P, mod.value.decode(), name.value.decode())
        assert a.shape == tuple(shape[d] for d in range(nd.value)), (mod.value, name.value)
        seen += 1
    assert seen == sum(len([k for k in v if k in ("w", "b", "embeddings", "scale", "offset")]) for v in P.values()) + 6
    # error paths: forward before pack, unknown array, wrong shape, bad sizes
    assert lib.vtts_nat_duration_forward(h, C.c_void_p(256), C.c_void_p(256), 1, 4, C.c_void_p(256), C.c_void_p(256), 1 << 30, None) == -2
    buf = (C.c_float * 4)()
    shp = (C.c_int64 * 3)(1, 2, 3)
    assert lib.vtts_nat_duration_set_param(h, b"nope", b"w", buf, shp, 3) == -1
    assert lib.vtts_nat_duration_set_param(h, b"linear", b"w", buf, shp, 2) == -6
    nb = C.c_size_t(0)
    assert lib.vtts_nat_duration_workspace_bytes(h, 0, 4, C.byref(nb)) == -1
    assert lib.vtts_nat_duration_pack(h, C.c_void_p(256), 1 << 30, None) == -3
    lib.vtts_nat_duration_destroy(h)
    bad = _lib.NatDurationCfg(256, 300)
    assert lib.vtts_nat_duration_create(C.byref(bad), 0, C.byref(h)) == -1


def test_oracle_fp32_vs_fp64_and_basic_properties():
    P, S = synthetic_duration_checkpoint()
    rng = np.random.default_rng(3)
    for L in (1, 2, 37, 120):
        tok = rng.integers(0, 100, size=L)
        d32 = no.duration_model(P, S, tok, dtype=np.float32)
        d64 = no.duration_model(P, S, tok, dtype=np.float64)
        assert d32.shape == (L,) and np.all(d32 > 0)  # softplus
        assert np.abs(d32 - d64).max() < 5e-6
    # the backward LSTM sees the future: changing the last token changes the first duration; the forward one does not
    # see it in its own half of the encoding
    tok = rng.integers(0, 100, size=30)
    tok2 = tok.copy()
    tok2[-1] = (tok2[-1] + 1) % 100
    assert no.duration_model(P, S, tok)[0] != no.duration_model(P, S, tok2)[0]


def test_text2mel_frame_rules_on_oracle_durations():
    """The integer quantities of text2mel.py:78-79, :99-101 from fp32 durations (oracle stands in for the network)."""
    from viettts_amd.nat import text2mel as t2m
    from viettts_amd.nat.config import FLAGS

    P, S = synthetic_duration_checkpoint()
    tokens = [FLAGS.sil_index, 10, 11, FLAGS.word_end_index, 12, FLAGS.word_end_index, FLAGS.sil_index]
    d = no.duration_model(P, S, np.array(tokens))[None, :]
    d = t2m.apply_duration_rules(tokens, d, 0.2)
    assert d[0, 3] == 0 and d[0, 5] == 0 and d[0, 0] >= 0.2 and d[0, -1] >= 0.2
    n = t2m.n_frames_from_durations(d)
    assert n == int(np.float32(np.sum((d * np.float32(16000)) / np.float32(256), dtype=np.float32)))
    assert t2m.trailing_silence_frames(d) == int(float(d[0, -1]) * 16000 / 256)


def test_acoustic_param_table_matches_synthetic_checkpoint(lib):
    from viettts_amd.nat.duration import _lookup
    from viettts_amd.nat.synth import synthetic_acoustic_checkpoint

    h = C.c_void_p(0)
    cfg = _lib.NatAcousticCfg(256, 256, 512, 256, 80, 512)
    _lib.check(lib, lib.vtts_nat_acoustic_create(C.byref(cfg), 0, C.byref(h)))
    n = C.c_int(0)
    _lib.check(lib, lib.vtts_nat_acoustic_num_params(h, C.byref(n)))
    P, S = synthetic_acoustic_checkpoint()
    for i in range(n.value):
        mod, name = C.c_char_p(), C.c_char_p()
        shape = (C.c_int64 * 3)()
        nd = C.c_int(0)
        _lib.check(lib, lib.vtts_nat_acoustic_param_info(h, i, C.byref(mod), C.byref(name), shape, C.byref(nd)))
        a = _lookup(S if name.value == b"average" else P, "acoustic_model/~/" + mod.value.decode(), name.value.decode())
        assert a.shape == tuple(shape[d] for d in range(nd.value)), (mod.value, name.value)
    n_ckpt = sum(len([k for k in v if k != "hidden" and k != "counter"]) for v in list(P.values()) + list(S.values()))
    assert n.value == n_ckpt
    nb = C.c_size_t(0)
    assert lib.vtts_nat_acoustic_workspace_bytes(h, 1, 10, 0, C.byref(nb)) == -1
    lib.vtts_nat_acoustic_destroy(h)
    bad = _lib.NatAcousticCfg(256, 256, 500, 256, 80, 512)  # decoder_dim not a multiple of 32 (matrix-core k-steps)
    assert lib.vtts_nat_acoustic_create(C.byref(bad), 0, C.byref(h)) == -1
    other = _lib.NatAcousticCfg(256, 128, 256, 64, 80, 256)  # other widths are fine
    assert lib.vtts_nat_acoustic_create(C.byref(other), 0, C.byref(h)) == 0
    lib.vtts_nat_acoustic_destroy(h)


def test_acoustic_oracle_properties():
    from viettts_amd.nat.synth import synthetic_acoustic_checkpoint

    P, S = synthetic_acoustic_checkpoint()
    tok = np.random.default_rng(0).integers(0, 100, size=6)
    dur = np.array([2.0, 3.5, 0.0, 4.0, 1.5, 3.0], np.float32)
    nf = int(dur.sum())
    m32 = no.acoustic_inference(P, S, tok, dur, nf, dtype=np.float32)
    m64 = no.acoustic_inference(P, S, tok, dur, nf, dtype=np.float64)
    assert m32.shape == (nf, 80) and np.abs(m32 - m64).max() < 1e-4
    # Gaussian upsampling rows are convex combinations of encoder rows (weights sum to 1)
    x = np.eye(6, dtype=np.float64)
    w = no.gaussian_upsample(x, dur.astype(np.float64), nf)
    assert np.allclose(w.sum(axis=1), 1.0) and w.min() >= 0


def test_gates_from_the_token_rows_identity():
    """What csrc/nat.hip's gate path relies on (round 4): the conditioning only ever meets the first E rows of the LSTMs' input matrices, and
    upsampling is linear, so  b + upsample(enc) @ W[0:E]  ==  b + upsample(enc @ W[0:E]) — the GEMM may run over the tokens' rows and the
    upsampling weights mix its rows into the frames'.  Exact in exact arithmetic; in fp32 the two orders differ by rounding only."""
    rng = np.random.default_rng(5)
    L, E, G4 = 23, 64, 96
    dur = np.abs(rng.normal(3.0, 1.5, size=L))
    dur[7] = 0.0
    nf = int(dur.sum())
    W, b = rng.normal(size=(E, G4)) / np.sqrt(E), rng.normal(size=G4) * 0.1
    enc = rng.normal(size=(L, E))
    by_frames = no.gaussian_upsample(enc, dur, nf) @ W + b
    by_tokens = no.gaussian_upsample(enc @ W, dur, nf) + b
    assert by_frames.shape == (nf, G4) and np.abs(by_frames - by_tokens).max() < 1e-12
    f32 = lambda a: a.astype(np.float32)
    d32 = np.abs((no.gaussian_upsample(f32(enc), f32(dur), nf) @ f32(W) + f32(b)) - (no.gaussian_upsample(f32(enc) @ f32(W), f32(dur), nf) + f32(b))).max()
    assert d32 < 2e-5, d32


def test_threefry_known_answers():
    """Random123's kat_vectors for threefry2x32 with 20 rounds (the same three jax's own test suite uses) pin the cipher
    behind the device-drawn dropout masks."""
    from oracle.nat_oracle import threefry2x32_20, threefry_keep_masks

    for key, ctr, want in (
        ((0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6B200159, 0x99BA4EFE)),
        ((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0x1CB996FC, 0xBB002BE7)),
        ((0x13198A2E, 0x03707344), (0x243F6A88, 0x85A308D3), (0xC4923A9C, 0x483DF7A0)),
    ):
        x0, x1 = threefry2x32_20(key[0], key[1], ctr[0], ctr[1])
        assert (int(x0), int(x1)) == want
    m = threefry_keep_masks(12345, 40, 256)
    assert m.shape == (40, 2, 256) and 0.47 < m.mean() < 0.53
    assert not np.array_equal(m, threefry_keep_masks(12346, 40, 256))
    assert np.array_equal(m[:10], threefry_keep_masks(12345, 10, 256))  # a frame's mask does not depend on the sentence's length


def test_jax_legacy_prng_known_answers_and_haiku_mask_schedule():
    """The reference's OWN mask stream (oracle/nat_oracle.py::haiku_prenet_keep_masks): jax.random's classic threefry layout,
    pinned by the two values JAX's documentation prints for ``PRNGKey(0)`` — ``random.split(key)`` and ``random.uniform(key)``
    (docs "Pseudo random numbers in JAX" / the jax.random module docstring) — on top of Random123's cipher vectors above."""
    from oracle.nat_oracle import haiku_prenet_keep_masks, jax_legacy_split, jax_legacy_uniform

    k0 = np.array([0, 0], dtype=np.uint32)  # jax.random.PRNGKey(0)
    assert jax_legacy_split(k0).tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
    u = jax_legacy_uniform(k0, 1)
    assert u.dtype == np.float32 and abs(float(u[0]) - 0.41845703) < 5e-9
    # odd / even word counts: the first words of a longer draw are NOT those of a shorter one (the counters are halved)
    assert jax_legacy_uniform(k0, 4)[0] != jax_legacy_uniform(k0, 2)[0]
    rng = np.array([123456789, 42], dtype=np.uint32)
    m = haiku_prenet_keep_masks(rng, 30, 256)
    assert m.shape == (30, 2, 256) and 0.47 < m.mean() < 0.53
    assert np.array_equal(m[:7], haiku_prenet_keep_masks(rng, 7, 256))  # frame f's masks depend on the key chain up to f only
    # the chain: frame 1's first mask comes from the subkey of the THIRD split
    key = rng
    for _ in range(3):
        key, sub = jax_legacy_split(key, 2)
    assert np.array_equal(m[1, 0], jax_legacy_uniform(sub, 256) < np.float32(0.5))
    assert not np.array_equal(m[0, 0], m[0, 1])


# ---- the reference's own NAT code, executed (oracle/make_nat_golden.py: vietTTS/nat/text2mel.py + model.py from /root/reference under
# ---- oracle/haiku_shim.py) -------------------------------------------------------------------------------------------------------
NAT_GOLDEN = Path(__file__).parent / "golden" / "nat_text2mel_golden.npz"


def _synthetic_checkpoints_checked(g):
    from oracle.make_nat_golden import params_digest
    from viettts_amd.nat.synth import synthetic_acoustic_checkpoint

    dp, ds = synthetic_duration_checkpoint()
    ap, as_ = synthetic_acoustic_checkpoint()
    assert params_digest(dp, ds, ap, as_) == str(g["params_sha256"]), "the seeded synthetic checkpoints differ from the ones the fixture was minted with"
    return dp, ds, ap, as_


def test_oracle_equals_the_reference_code_executed():
    """oracle/nat_oracle.py (the restatement every GPU test is checked against) against the output of the reference's own
    text2mel.py / model.py: durations bit for bit, mel to fp64 rounding — the wiring, the skip-connection order, the order of rng
    draws behind the always-on prenet dropout and the frame arithmetic are the reference's."""
    g = np.load(NAT_GOLDEN)
    dp, ds, ap, as_ = _synthetic_checkpoints_checked(g)
    for ci in range(2):
        p = f"c{ci}_"
        tok = g[p + "tokens"]
        assert [int(t) for t in tok] == t2m.text2tokens(str(g[p + "text"]), Path(__file__).parent / "golden" / "text" / "lexicon.txt")
        d = no.duration_model(dp, ds, tok, dtype=np.float64)
        assert np.abs(d - g[p + "durations_s"][0]).max() < 1e-13
        ruled = t2m.apply_duration_rules([int(t) for t in tok], g[p + "durations_s"], float(g[p + "silence_duration"]))
        assert np.abs(np.asarray(ruled, np.float64) - g[p + "durations_ruled_s"]).max() < 1e-7  # the product's rules run in float32
        nf = int(g[p + "n_frames"])
        assert t2m.n_frames_from_durations(ruled) == nf and t2m.trailing_silence_frames(ruled) == int(g[p + "trailing_frames"])
        masks = no.haiku_prenet_keep_masks(g["rng_key"], nf)
        frames = g[p + "durations_ruled_s"][0] * FLAGS.sample_rate / (FLAGS.n_fft // 4)
        mel = no.acoustic_inference(ap, as_, tok, frames, nf, prenet_masks=lambda t: (masks[t, 0], masks[t, 1]), dtype=np.float64)
        assert np.abs(mel - g[p + "mel_full"]).max() < 1e-11


@pytest.mark.skipif(not Path("/root/reference/vietTTS/nat/model.py").exists(), reason="needs /root/reference (build container only)")
def test_reference_nat_code_reproduces_the_committed_fixture(tmp_path):
    """Re-mint the fixture from the reference's files and compare with the committed one (where the reference exists)."""
    import os
    import subprocess
    import sys

    out = tmp_path / "again.npz"
    env = dict(os.environ, VTTS_NAT_GOLDEN_OUT=str(out))
    r = subprocess.run([sys.executable, str(Path(__file__).parents[1] / "oracle" / "make_nat_golden.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    a, b = np.load(NAT_GOLDEN), np.load(out)
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


def test_checkpoint_with_haiku_and_jax_objects_loads_without_those_libraries(tmp_path):
    """A checkpoint pickled the way the reference pickles it (vietTTS/nat/utils.py:18-26: Haiku mappings of jax arrays, an optax state)
    read where none of those libraries exists.  The classes are EMULATED here under their real module paths (pickling hooks restated
    from the libraries: FlatMapping reduces to its plain mapping, a jax array to numpy's reduce tuple + an aval state) and removed from
    ``sys.modules`` before loading; no real checkpoint is available offline to confirm the formats (viettts_amd/nat/ckpt.py)."""
    import pickle
    import sys
    import types
    from collections import namedtuple

    from viettts_amd.nat.ckpt import CheckpointFormatError, load_checkpoint

    fake = {}

    def mod(name):
        m = types.ModuleType(name)
        fake[name] = m
        sys.modules[name] = m
        return m

    for n in ("haiku", "haiku._src", "jax", "jax._src", "optax", "optax._src"):
        mod(n)
    ds, arr, tr = mod("haiku._src.data_structures"), mod("jax._src.array"), mod("optax._src.transform")

    class FlatMapping(dict):
        def __reduce__(self):
            return (FlatMapping, (dict(self),))

    FlatMapping.__module__, FlatMapping.__qualname__ = ds.__name__, "FlatMapping"
    ds.FlatMapping = FlatMapping

    def _reconstruct_array(fun, args, arr_state, aval_state):
        raise AssertionError("the real constructor must not be needed")

    _reconstruct_array.__module__, _reconstruct_array.__qualname__ = arr.__name__, "_reconstruct_array"
    arr._reconstruct_array = _reconstruct_array

    class ArrayImpl:
        def __init__(self, v):
            self.v = np.asarray(v)

        def __reduce__(self):
            fun, args, state = self.v.__reduce__()
            return (_reconstruct_array, (fun, args, state, {"weak_type": False, "named_shape": {}}))

    ArrayImpl.__module__, ArrayImpl.__qualname__ = arr.__name__, "ArrayImpl"
    arr.ArrayImpl = ArrayImpl
    ScaleByAdamState = namedtuple("ScaleByAdamState", ["count", "mu", "nu"])
    ScaleByAdamState.__module__, ScaleByAdamState.__qualname__ = tr.__name__, "ScaleByAdamState"
    tr.ScaleByAdamState = ScaleByAdamState

    P, S = synthetic_duration_checkpoint()
    wrap = lambda d: FlatMapping({k: FlatMapping({n: ArrayImpl(a) for n, a in v.items()}) for k, v in d.items()})
    dic = {"step": 7, "params": wrap(P), "aux": wrap(S), "rng": ArrayImpl(np.array([1, 2], np.uint32)),
           "optim_state": (ScaleByAdamState(ArrayImpl(np.zeros((), np.int32)), wrap(P), wrap(P)),)}
    path = tmp_path / "duration_latest_ckpt.pickle"
    try:
        with open(path, "wb") as f:
            pickle.dump(dic, f)
    finally:
        for n in fake:
            del sys.modules[n]
    with pytest.raises(ModuleNotFoundError):
        with open(path, "rb") as f:
            pickle.load(f)  # what the reference's loader does: needs haiku / jax / optax
    got = load_checkpoint(path)
    assert got["step"] == 7 and sorted(got["params"]) == sorted(P) and sorted(got["aux"]) == sorted(S)
    for k in P:
        for n in P[k]:
            assert isinstance(got["params"][k][n], np.ndarray) and np.array_equal(got["params"][k][n], P[k][n])
    assert np.array_equal(got["rng"], np.array([1, 2], np.uint32))
    params, state = t2m.load_duration_checkpoint(path)
    assert np.array_equal(params["duration_model/~/linear_1"]["b"], P["duration_model/~/linear_1"]["b"]) and sorted(state) == sorted(S)
    # a layout it cannot recognise is refused by name, not guessed
    from viettts_amd.nat import ckpt as ck

    weird = ck._absent_class("haiku._src.data_structures", "Mystery")({"a": 1}, {"b": 2})
    with pytest.raises(CheckpointFormatError, match="Mystery"):
        ck.to_plain(weird)


def test_partitionable_threefry_layout_is_another_stream_with_the_same_schedule():
    """jax_threefry_partitionable (JAX >= 0.5's default): an UNPINNED restatement (oracle/nat_oracle.py: no known answer for this mode is
    quotable offline) — checked for what can be checked here: its defining property (element i of a draw and subkey i of a split depend
    on i alone, not on the size of the draw: that is what "partitionable" means), the Haiku schedule on top of it, and that it is not the
    classic stream."""
    from oracle.nat_oracle import haiku_prenet_keep_masks, jax_legacy_uniform, jax_partitionable_split, jax_partitionable_uniform, threefry2x32_20

    k = np.array([123456789, 42], dtype=np.uint32)
    assert np.array_equal(jax_partitionable_uniform(k, 4)[:2], jax_partitionable_uniform(k, 2))  # prefix-stable: the classic layout is not
    assert np.array_equal(jax_partitionable_split(k, 4)[:2], jax_partitionable_split(k, 2))
    y0, y1 = threefry2x32_20(np.uint32(k[0]), np.uint32(k[1]), np.zeros(1, np.uint32), np.ones(1, np.uint32))
    assert jax_partitionable_split(k, 2)[1].tolist() == [int(y0[0]), int(y1[0])]
    assert not np.array_equal(jax_partitionable_uniform(k, 8), jax_legacy_uniform(k, 8))
    m = haiku_prenet_keep_masks(k, 20, 256, partitionable=True)
    assert m.shape == (20, 2, 256) and 0.47 < m.mean() < 0.53
    assert np.array_equal(m[:5], haiku_prenet_keep_masks(k, 5, 256, partitionable=True))
    key = k
    for _ in range(3):
        key, sub = jax_partitionable_split(key, 2)
    assert np.array_equal(m[1, 0], jax_partitionable_uniform(sub, 256) < np.float32(0.5))
