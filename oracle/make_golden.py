"""ORACLE tooling — generates tests/golden/*.npz from the REAL reference, in the build container.

Runs only where /root/reference exists (it does not on the GPU box; nothing at test/bench
time imports this).  What it does:

1. draws the synthetic weight sets of viettts_amd.hifigan.synth (seeded);
2. builds the reference's own PyTorch generator (vietTTS/hifigan/torch_model.py:156-218),
   loads the weights as a weight-norm checkpoint (weight_v = W, weight_g = ||W||, so that
   remove_weight_norm() reproduces W bit-exactly), saves it as an upstream ``g_*`` file;
3. runs the reference's own converter on that file
   (vietTTS/hifigan/convert_torch_model_to_haiku.py:27-62) and checks our layout map
   (viettts_amd.hifigan.weights.state_dict_to_haiku) reproduces its ``hk_hifi.pickle``
   bit-for-bit;
4. runs the reference generator forward in fp32 and fp64 on the synthetic mels and stores
   the outputs (small shapes in full; the BASELINE shapes T=512 and 64 x 1024 — three rows of the benchmark's own
   batch — as strided samples + sums).

Usage:  python oracle/make_golden.py [--only CASE ...]
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import pickle
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
sys.path.insert(0, str(REPO))

from viettts_amd.hifigan.config import TINY, TINY2, V1, HifiganConfig  # noqa: E402
from viettts_amd.hifigan.synth import params_digest, synthetic_mel, synthetic_params  # noqa: E402
from viettts_amd.hifigan.weights import conv_specs, haiku_to_state_dict, state_dict_to_haiku  # noqa: E402


class AttrDict(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.__dict__ = self


def _import_reference():
    """Import the reference's hifigan modules from where they lie (read-only)."""
    sys.path.insert(0, str(REF))
    import vietTTS.hifigan.torch_model as tm  # noqa: E402
    import vietTTS.hifigan.convert_torch_model_to_haiku as conv  # noqa: E402

    return tm, conv


def cfg_to_h(cfg: HifiganConfig) -> AttrDict:
    return AttrDict(
        resblock=cfg.resblock,
        upsample_rates=list(cfg.upsample_rates),
        upsample_kernel_sizes=list(cfg.upsample_kernel_sizes),
        upsample_initial_channel=cfg.upsample_initial_channel,
        resblock_kernel_sizes=list(cfg.resblock_kernel_sizes),
        resblock_dilation_sizes=[list(d) for d in cfg.resblock_dilation_sizes],
    )


def reference_generator(tm, cfg, params, dtype):
    """Reference torch generator holding ``params`` (Haiku dict) — weight norm removed."""
    with contextlib.redirect_stdout(io.StringIO()):
        g = tm.Generator(cfg_to_h(cfg))
        g.remove_weight_norm()
    sd = {k: torch.from_numpy(v) for k, v in haiku_to_state_dict(cfg, params).items()}
    missing = g.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return g.eval().to(dtype)


def reference_forward(g, mel_nwc: np.ndarray, dtype):
    """mel [B,T,80] NWC -> reference output [B, 256T] and pre-tanh (via a hook)."""
    pre = {}
    hnd = g.conv_post.register_forward_hook(lambda m, i, o: pre.__setitem__("x", o.detach()))
    with torch.no_grad():
        x = torch.from_numpy(mel_nwc).to(dtype).permute(0, 2, 1).contiguous()  # oracle wants NCW
        y = g(x)
    hnd.remove()
    return y[:, 0].numpy(), pre["x"][:, 0].numpy()


def check_converter(tm, conv, cfg, params, workdir: Path):
    """Step 2+3: our layout map must equal the reference converter's pickle bit-for-bit."""
    with contextlib.redirect_stdout(io.StringIO()):
        g = tm.Generator(cfg_to_h(cfg))  # with weight norm
    sd_plain = haiku_to_state_dict(cfg, params)
    sd = g.state_dict()
    for spec in conv_specs(cfg):
        w = torch.from_numpy(sd_plain[spec.torch_prefix + ".weight"])
        sd[spec.torch_prefix + ".weight_v"] = w
        sd[spec.torch_prefix + ".weight_g"] = torch.norm_except_dim(w, 2, 0)
        sd[spec.torch_prefix + ".bias"] = torch.from_numpy(sd_plain[spec.torch_prefix + ".bias"])
    ckpt = workdir / "g_00000000"
    torch.save({"generator": sd}, ckpt)
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        a = types.SimpleNamespace(checkpoint_file=str(ckpt))
        with contextlib.redirect_stdout(io.StringIO()):
            conv.convert_to_haiku(a, cfg_to_h(cfg), torch.device("cpu"))
        with open(workdir / "assets/infore/hifigan/hk_hifi.pickle", "rb") as f:
            ref_hk = pickle.load(f)
    finally:
        os.chdir(cwd)
    assert set(ref_hk) == set(params), (sorted(set(ref_hk) ^ set(params))[:5])
    worst = 0.0
    for k in params:
        for n in ("w", "b"):
            r = np.asarray(ref_hk[k][n])
            assert r.shape == params[k][n].shape, (k, n, r.shape, params[k][n].shape)
            worst = max(worst, float(np.abs(r - params[k][n]).max()))
    # `worst` is bounded by the 1-ulp rounding of torch's weight-norm fold v*(g/||v||).  The
    # LAYOUT map is checked bit-exactly: fold with the reference's own remove_weight_norm(),
    # map its state dict with our function, compare with the reference converter's pickle.
    with contextlib.redirect_stdout(io.StringIO()):
        g2 = tm.Generator(cfg_to_h(cfg))
        g2.load_state_dict(sd)
        g2.remove_weight_norm()
    ours = state_dict_to_haiku(cfg, {k: v.numpy() for k, v in g2.state_dict().items()})
    exact = max(float(np.abs(ours[k][n] - np.asarray(ref_hk[k][n])).max()) for k in params for n in ("w", "b"))
    # and our own reader of weight-norm checkpoints (fp64 norm) agrees to rounding
    ours_wn = state_dict_to_haiku(cfg, {k: v.numpy() for k, v in sd.items()})
    worst2 = max(float(np.abs(ours_wn[k][n] - np.asarray(ref_hk[k][n])).max()) for k in params for n in ("w", "b"))
    return worst, exact, worst2, ref_hk


def reference_haiku_mel2wave(cfg, params, mel):
    """The reference's HAIKU generator, executed: ``vietTTS/hifigan/mel2wave.py::mel2wave`` (with ``model.py``'s Generator /
    ResBlock1 / ResBlock2) imported from /root/reference and run unchanged — config from ``assets/hifigan/config.json`` and
    weights from ``./assets/infore/hifigan/hk_hifi.pickle`` under a scratch CWD, as it reads them — over oracle/haiku_shim.py
    (jax / haiku cannot be installed here), in float64.  The shim's convolutions are oracle/hifigan_oracle.py's; what this run
    adds is the Haiku model's own wiring and module names (the pickle keys are whatever model.py's modules ask for)."""
    from oracle import haiku_shim as shim

    shim.install()
    shim.set_dtype(np.float64)
    import vietTTS.hifigan.mel2wave as ref_m2w  # noqa: E402  (the reference's file, unchanged; `vietTTS` is already the reference's package)

    assert Path(ref_m2w.__file__).resolve().is_relative_to(REF), ref_m2w.__file__
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        (td / "assets/hifigan").mkdir(parents=True)
        (td / "assets/infore/hifigan").mkdir(parents=True)
        with open(td / "assets/hifigan/config.json", "w") as f:
            json.dump(dict(cfg_to_h(cfg)), f)
        with open(td / "assets/infore/hifigan/hk_hifi.pickle", "wb") as f:
            pickle.dump({k: {n: np.asarray(a) for n, a in m.items()} for k, m in params.items()}, f)
        cwd = os.getcwd()
        os.chdir(td)
        try:
            return np.asarray(ref_m2w.mel2wave(np.asarray(mel, dtype=np.float64)))
        finally:
            os.chdir(cwd)


def mint_converter_golden(tm, conv, out: Path):
    """A weight-norm checkpoint of the TINY architecture (g != ||v||, so the fold matters) and what the REFERENCE converter
    (convert_torch_model_to_haiku.py:27-62) writes for it: tests/golden/convert_tiny.npz — pins the converter entry point
    vietTTS.hifigan.convert_torch_model_to_haiku of this repo bit-for-bit (tests/test_weights.py)."""
    cfg = TINY
    with contextlib.redirect_stdout(io.StringIO()):
        g = tm.Generator(cfg_to_h(cfg))
    gen = torch.Generator().manual_seed(2468)
    sd = g.state_dict()
    for spec in conv_specs(cfg):
        v = sd[spec.torch_prefix + ".weight_v"]
        sd[spec.torch_prefix + ".weight_v"] = torch.randn(v.shape, generator=gen) * 0.3
        sd[spec.torch_prefix + ".weight_g"] = torch.rand(sd[spec.torch_prefix + ".weight_g"].shape, generator=gen) + 0.5
        sd[spec.torch_prefix + ".bias"] = torch.randn(sd[spec.torch_prefix + ".bias"].shape, generator=gen) * 0.1
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        torch.save({"generator": sd}, td / "g_00000001")
        cwd = os.getcwd()
        os.chdir(td)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                conv.convert_to_haiku(types.SimpleNamespace(checkpoint_file=str(td / "g_00000001")), cfg_to_h(cfg), torch.device("cpu"))
            with open(td / "assets/infore/hifigan/hk_hifi.pickle", "rb") as f:
                ref_hk = pickle.load(f)
        finally:
            os.chdir(cwd)
    arrs = {}
    for k, v in sd.items():
        arrs["SD::" + k] = v.numpy()
    for k, mod in ref_hk.items():
        for n, a in mod.items():
            arrs["HK::" + k + "::" + n] = np.ascontiguousarray(a)
    np.savez_compressed(out / "convert_tiny.npz", **arrs)
    print(f"[convert_tiny] {len(sd)} checkpoint arrays -> {len(ref_hk)} Haiku modules (reference converter)")


def main():
    tm, conv = _import_reference()
    out = REPO / "tests" / "golden"
    out.mkdir(parents=True, exist_ok=True)
    meta = {"torch": torch.__version__, "numpy": np.__version__, "cases": {}}

    cases = [
        # name, cfg, weight kind, wseed, B, T, mel seed, store-full?
        ("tiny_scaled_T12", TINY, "scaled", 4321, 2, 12, 1234, True),
        # ResBlock2 generator (model.py:54-74; torch_model.py:117-148): the reference builds it, its converter cannot name it
        ("tiny2_scaled_T12", TINY2, "scaled", 4321, 2, 12, 1234, True),
        ("v1_scaled_T8", V1, "scaled", 4321, 1, 8, 1234, True),
        ("v1_scaled_T37", V1, "scaled", 4321, 2, 37, 77, True),
        ("v1_init_T16", V1, "init", 1234, 1, 16, 1234, True),
        ("v1_scaled_T512", V1, "scaled", 4321, 1, 512, 1234, False),
        # BASELINE configs[2]'s own batch (bench.py rank 0: synthetic_mel(64, 1024, 1234)); rows 0, 37, 63 of it
        ("v1_scaled_B64_T1024", V1, "scaled", 4321, 64, 1024, 1234, False),
    ]
    rows_of = {"v1_scaled_B64_T1024": [0, 37, 63]}
    only = sys.argv[sys.argv.index("--only") + 1:] if "--only" in sys.argv else None
    if only:
        with open(out / "golden_meta.json") as f:
            meta = json.load(f)
    if not only or "convert_tiny" in only:
        mint_converter_golden(tm, conv, out)
    conv_checked = set()
    for name, cfg, kind, wseed, B, T, mseed, full in cases:
        if only and name not in only:
            continue
        params = synthetic_params(cfg, wseed, kind)
        digest = params_digest(params)
        ck = (id(cfg), kind, wseed)
        if cfg.resblock == "2":
            # the reference converter writes res_block2_N/~/convs_Z (convert_torch_model_to_haiku.py:45-46), names its own Haiku
            # model never creates (model.py:105: res_block1_N with default-named hk.Conv1D): nothing to compare layouts with
            conv_checked.add(ck)
        if ck not in conv_checked:
            with tempfile.TemporaryDirectory() as td:
                w1, ex, w2, ref_hk = check_converter(tm, conv, cfg, params, Path(td))
            print(f"[{name}] reference converter vs ours: layout map max|diff| = {ex} (must be 0); "
                  f"weight-norm fold rounding {w1:.2e}; our weight-norm reader {w2:.2e}")
            assert ex == 0.0, "layout map differs from the reference converter"
            assert w1 < 1e-6 and w2 < 1e-6
            conv_checked.add(ck)
        mel = synthetic_mel(B, T, mseed, cfg.num_mels)
        rows = rows_of.get(name)
        if rows is not None:
            mel = np.ascontiguousarray(mel[rows])
        g32 = reference_generator(tm, cfg, params, torch.float32)
        y32, p32 = reference_forward(g32, mel, torch.float32)
        g64 = reference_generator(tm, cfg, params, torch.float64)
        y64, p64 = reference_forward(g64, mel, torch.float64)
        print(f"[{name}] ref fp32 vs fp64: max|dy| = {np.abs(y32 - y64).max():.3e}  max|dpre| = {np.abs(p32 - p64).max():.3e}  "
              f"|pre| max {np.abs(p64).max():.3f}  sat(|y|>0.99) {(np.abs(y64) > 0.99).mean():.3f}")
        rec = {"cfg": "TINY" if cfg is TINY else ("TINY2" if cfg is TINY2 else "V1"), "kind": kind, "wseed": wseed, "B": B, "T": T, "mseed": mseed,
               "params_sha256": digest}
        if rows is not None:
            rec["rows"] = rows
        arrs = {}
        if cfg is TINY or cfg is TINY2:
            # weights small enough to commit: fixture independent of the RNG
            for k, m in params.items():
                arrs["W::" + k + "::w"] = m["w"]
                arrs["W::" + k + "::b"] = m["b"]
            arrs["mel"] = mel
        if full:
            arrs.update(y32=y32.astype(np.float32), y64=y64.astype(np.float64), pre32=p32.astype(np.float32), pre64=p64.astype(np.float64))
            # the reference's OTHER implementation of the same generator — the Haiku one the product replaces — executed over the shim
            yhk = reference_haiku_mel2wave(cfg, params, mel)
            d = float(np.abs(yhk - np.squeeze(y64)).max())
            print(f"[{name}] reference Haiku mel2wave (model.py + mel2wave.py over oracle/haiku_shim.py) vs reference torch generator, fp64: max|dy| = {d:.3e}")
            assert yhk.shape == np.squeeze(y64).shape and d < 1e-12
            arrs["y64_haiku"] = yhk.astype(np.float64)
            rec["haiku_vs_torch_maxabs"] = d
        else:
            idx = np.arange(0, y64.shape[1], 61)
            if name == "v1_scaled_T512":  # BASELINE configs[1]'s own shape: the Haiku generator executed there too (B = 1)
                yhk = reference_haiku_mel2wave(cfg, params, mel)
                d = float(np.abs(yhk - np.squeeze(y64)).max())
                print(f"[{name}] reference Haiku mel2wave vs reference torch generator, fp64: max|dy| = {d:.3e}")
                assert yhk.shape == np.squeeze(y64).shape and d < 1e-12
                arrs["y64_haiku"] = yhk[idx].astype(np.float64)
                rec["haiku_vs_torch_maxabs"] = d
            arrs.update(idx=idx, y32=y32[:, idx].astype(np.float32), y64=y64[:, idx], pre32=p32[:, idx].astype(np.float32), pre64=p64[:, idx],
                        sum_y64=np.array([y64.sum(), np.abs(y64).sum(), (y64 ** 2).sum()]),
                        sum_pre64=np.array([p64.sum(), np.abs(p64).sum(), (p64 ** 2).sum()]))
        np.savez_compressed(out / f"{name}.npz", **arrs)
        meta["cases"][name] = rec
    with open(out / "golden_meta.json", "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", sorted(p.name for p in out.iterdir()))


if __name__ == "__main__":
    main()
