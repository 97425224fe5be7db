"""The oracle (oracle/hifigan_oracle.py) pinned against outputs of the reference's own
PyTorch generator (tests/golden/*.npz, minted by oracle/make_golden.py)."""
import json

import numpy as np
import pytest

from oracle.hifigan_oracle import (conv1d, conv1d_transpose, conv_transpose_same_pads, flops_per_frame,
                                   generator_forward, get_padding, mel2wave_oracle)
from viettts_amd.hifigan.config import TINY, TINY2, V1
from viettts_amd.hifigan.synth import params_digest, synthetic_mel, synthetic_params

CFG = {"V1": V1, "TINY": TINY, "TINY2": TINY2}


def _meta(golden_dir):
    with open(golden_dir / "golden_meta.json") as f:
        return json.load(f)


def _tiny_params_from_fixture(g):
    params = {}
    for name in g.files:
        if name.startswith("W::"):
            _, key, which = name.split("::")
            params.setdefault(key, {})[which] = g[name]
    return params


@pytest.mark.parametrize("case", ["tiny_scaled_T12", "tiny2_scaled_T12", "v1_scaled_T8", "v1_scaled_T37", "v1_init_T16"])
def test_oracle_matches_reference_full(golden_dir, case):
    rec = _meta(golden_dir)["cases"][case]
    cfg = CFG[rec["cfg"]]
    g = np.load(golden_dir / f"{case}.npz")
    params = synthetic_params(cfg, rec["wseed"], rec["kind"])
    # the seeded weights are the ones the golden was minted with
    assert params_digest(params) == rec["params_sha256"]
    mel = synthetic_mel(rec["B"], rec["T"], rec["mseed"], cfg.num_mels)
    y, pre = generator_forward(params, mel, cfg, np.float64, return_pre_tanh=True)
    assert y.shape == (rec["B"], cfg.hop * rec["T"], 1)
    # fp64 restatement of the Haiku semantics == fp64 reference torch generator
    assert np.abs(y[..., 0] - g["y64"]).max() < 1e-12
    assert np.abs(pre[..., 0] - g["pre64"]).max() < 1e-12
    # ... == the reference's HAIKU generator (vietTTS/hifigan/model.py + mel2wave.py from /root/reference, executed over
    # oracle/haiku_shim.py by oracle/make_golden.py::reference_haiku_mel2wave): the module the product replaces, its own wiring and names
    assert np.abs(np.squeeze(y[..., 0]) - g["y64_haiku"]).max() < 1e-12
    assert np.abs(np.squeeze(g["y64"]) - g["y64_haiku"]).max() < 1e-12 and rec["haiku_vs_torch_maxabs"] < 1e-12
    # and the reference's fp32 run sits at fp32 round-off from it
    assert np.abs(y[..., 0] - g["y32"]).max() < 2e-5
    # fp32 mode of the oracle (what the cpu_baseline leg times) is equally close
    y32 = generator_forward(params, mel, cfg, np.float32)
    assert y32.dtype == np.float32
    assert np.abs(y32[..., 0] - g["y64"]).max() < 2e-5


def test_tiny_fixture_is_rng_independent(golden_dir):
    """tiny_scaled_T12 carries its own weights and mel: parity does not hinge on torch's RNG."""
    g = np.load(golden_dir / "tiny_scaled_T12.npz")
    params = _tiny_params_from_fixture(g)
    y, pre = generator_forward(params, g["mel"], TINY, np.float64, return_pre_tanh=True)
    assert np.abs(y[..., 0] - g["y64"]).max() < 1e-12
    assert np.abs(pre[..., 0] - g["pre64"]).max() < 1e-12


def test_resblock2_fixture_is_rng_independent(golden_dir):
    """ResBlock2 (model.py:54-74) against the reference's torch generator built with "resblock": "2", on fixture weights."""
    g = np.load(golden_dir / "tiny2_scaled_T12.npz")
    params = _tiny_params_from_fixture(g)
    assert sum(1 for k in params if "res_block1_" in k) == 24  # 12 blocks x 2 convolutions
    y, pre = generator_forward(params, g["mel"], TINY2, np.float64, return_pre_tanh=True)
    assert np.abs(y[..., 0] - g["y64"]).max() < 1e-12
    assert np.abs(pre[..., 0] - g["pre64"]).max() < 1e-12


def test_oracle_matches_reference_baseline_shape(golden_dir):
    """BASELINE config 2 shape (B=1, T=512): strided sample + whole-tensor sums."""
    rec = _meta(golden_dir)["cases"]["v1_scaled_T512"]
    g = np.load(golden_dir / "v1_scaled_T512.npz")
    params = synthetic_params(V1, rec["wseed"], rec["kind"])
    assert params_digest(params) == rec["params_sha256"]
    mel = synthetic_mel(1, 512, rec["mseed"])
    y, pre = generator_forward(params, mel, V1, np.float32, return_pre_tanh=True)
    idx = g["idx"]
    assert np.abs(y[:, idx, 0] - g["y64"]).max() < 2e-5
    assert np.abs(pre[:, idx, 0] - g["pre64"]).max() < 2e-5
    assert np.abs(g["y64"][0] - g["y64_haiku"]).max() < 1e-12  # the reference's Haiku generator executed == its torch generator, here too
    s = g["sum_y64"]
    y64 = y[..., 0].astype(np.float64)
    assert abs(np.abs(y64).sum() - s[1]) / s[1] < 1e-5
    assert abs((y64 ** 2).sum() - s[2]) / s[2] < 1e-5


def test_mel2wave_oracle_contract():
    params = synthetic_params(TINY)
    mel = synthetic_mel(1, 5, 3, TINY.num_mels)
    w = mel2wave_oracle(params, mel, TINY)
    assert w.dtype == np.float32 and w.shape == (5 * 256,)
    assert np.all(np.abs(w) < 1.0)


def test_padding_rules():
    assert get_padding(3, 1) == 1 and get_padding(7, 3) == 9 and get_padding(11, 5) == 25
    assert conv_transpose_same_pads(16, 8) == (11, 11)
    assert conv_transpose_same_pads(4, 2) == (2, 2)


def test_flops_per_frame():
    assert flops_per_frame(V1) == 614105088  # 2 398 848 FLOP/sample * 256


def test_conv_primitives_against_torch():
    """Independent check of the two primitives against torch's CPU ops (A.1 / A.2 rules)."""
    import torch
    import torch.nn.functional as F

    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 19, 6))
    w = rng.standard_normal((7, 6, 5))
    b = rng.standard_normal(5)
    for d in (1, 3, 5):
        y = conv1d(x, w, b, d, get_padding(7, d))
        yt = F.conv1d(torch.from_numpy(x).permute(0, 2, 1), torch.from_numpy(w).permute(2, 1, 0), torch.from_numpy(b),
                      dilation=d, padding=get_padding(7, d)).permute(0, 2, 1).numpy()
        assert np.abs(y - yt).max() < 1e-12
    for k, s in ((16, 8), (4, 2)):
        wt = rng.standard_normal((k, 5, 6))  # haiku [K, Cout, Cin]
        y = conv1d_transpose(x, wt, b, s)
        w_torch = torch.from_numpy(np.ascontiguousarray(np.transpose(wt, (2, 1, 0))[:, :, ::-1]))  # [Cin,Cout,K] flipped
        yt = F.conv_transpose1d(torch.from_numpy(x).permute(0, 2, 1), w_torch, torch.from_numpy(b), stride=s,
                                padding=(k - s) // 2).permute(0, 2, 1).numpy()
        assert y.shape == (2, 19 * s, 5)
        assert np.abs(y - yt).max() < 1e-12


def test_built_reference_archive_matches_golden(golden_dir):
    """oracle/_ref/torch_generator_v1.pt — the reference's own torch generator compiled by oracle/build_ref.py, the thing
    bench.py's cpu_baseline times — reproduces the committed golden vectors (minted from the eager reference)."""
    import torch

    from oracle.build_ref import load_reference_archive
    from viettts_amd.hifigan.weights import haiku_to_state_dict

    rec = _meta(golden_dir)["cases"]["v1_scaled_T37"]
    g = np.load(golden_dir / "v1_scaled_T37.npz")
    sd = {k: torch.from_numpy(v) for k, v in haiku_to_state_dict(V1, synthetic_params(V1, rec["wseed"], rec["kind"])).items()}
    ts = load_reference_archive(sd)
    if ts is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box and no prebuilt archive)")
    mel = synthetic_mel(rec["B"], rec["T"], rec["mseed"])
    with torch.no_grad():
        y = ts(torch.from_numpy(mel).permute(0, 2, 1).contiguous())[:, 0].numpy()
    assert np.abs(y - g["y32"]).max() < 1e-6  # same graph, same weights (thread-count reassociation only)
    assert np.abs(y - g["y64"]).max() < 2e-5


@pytest.mark.skipif(not __import__("pathlib").Path("/root/reference/vietTTS/hifigan/model.py").exists(), reason="needs /root/reference (build container only)")
def test_reference_haiku_generator_reproduces_the_committed_fixture(golden_dir):
    """Run the reference's Haiku mel2wave again (over the shim, in a subprocess: it installs stand-in `jax` / `haiku` modules) and
    compare with the committed `y64_haiku` of the TINY fixture, whose weights and mel are in the file."""
    import subprocess
    import sys
    import textwrap

    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        from oracle import make_golden as mg
        mg._import_reference()
        g = np.load(%r)
        params = {}
        for name in g.files:
            if name.startswith("W::"):
                _, key, which = name.split("::")
                params.setdefault(key, {})[which] = g[name]
        y = mg.reference_haiku_mel2wave(mg.TINY, params, g["mel"])
        assert np.array_equal(y, g["y64_haiku"]), float(np.abs(y - g["y64_haiku"]).max())
        print("same")
    """) % (str(golden_dir.parents[1]), str(golden_dir / "tiny_scaled_T12.npz"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "same" in r.stdout, r.stdout + r.stderr


def test_bench_parity_grade_fixture_is_what_the_oracle_gives(golden_dir):
    """tests/golden/bench_parity_grade.npz (minted by oracle/make_bench_golden.py; read by bench.py's parity-grade legs, which must not import the
    oracle): one long-form window re-minted here — the oracle generator on frames [lo - 13, hi + 13) of the 10-minute utterance's mel — equals the
    fixture exactly, and the fixture's pipeline sentences carry consistent integer frame counts."""
    import oracle.make_bench_golden as mk
    from oracle.hifigan_oracle import generator_forward
    from viettts_amd.hifigan.config import V1
    from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params

    g = np.load(golden_dir / "bench_parity_grade.npz")
    lo = int(g["lf_start_lo"])
    assert lo == 0 and g["lf_start_wave"].shape == (256 * mk.WIN,)
    mel = synthetic_mel(1, mk.T10, 99)
    y = generator_forward(synthetic_params(V1, 4321, "scaled"), mel[:, : mk.WIN + mk.HALO], V1, np.float64)[0, :, 0]
    assert np.array_equal(y[: 256 * mk.WIN], g["lf_start_wave"])
    assert int(g["lf_seam_lo"]) == mk.CHUNK * 37 - mk.WIN // 2 and int(g["lf_end_lo"]) == mk.T10 - mk.WIN
    for i in (int(v) for v in g["pipe_sentences"]):
        nfr, trail = (int(v) for v in g[f"pipe_{i}_frames"])
        assert g[f"pipe_{i}_wave"].shape == (256 * (nfr - trail),) and 0 <= trail < nfr
        assert np.isfinite(g[f"pipe_{i}_wave"]).all() and float(np.abs(g[f"pipe_{i}_wave"]).max()) < 1.0
