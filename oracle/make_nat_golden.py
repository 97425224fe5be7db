#!/usr/bin/env python
"""ORACLE — TEST INFRASTRUCTURE ONLY.  Mints tests/golden/nat_text2mel_golden.npz by EXECUTING THE REFERENCE'S OWN NAT CODE.

Runs only where /root/reference exists (this container): the reference's ``vietTTS/nat/text2mel.py`` (``text2mel``,
``predict_duration``, ``predict_mel``, ``text2tokens``) and ``vietTTS/nat/model.py`` (``DurationModel``,
``AcousticModel.inference``) are imported from where they lie and run unchanged, with ``oracle/haiku_shim.py`` standing in
for ``haiku`` / ``jax`` (not installable offline).  Checkpoints: the seeded synthetic ones of ``viettts_amd/nat/synth.py``,
written as ``duration_latest_ckpt.pickle`` / ``acoustic_latest_ckpt.pickle`` under a scratch ``assets/infore/nat`` exactly as
the reference reads them (text2mel.py:27-28, :62-71), with an ``rng`` key so that the always-on prenet dropout
(model.py:95-100) draws from the checkpoint's stream as in the reference.

What the fixture pins: the WIRING of the two networks, the order of rng draws, and text2mel's silence rules / frame arithmetic
to the reference's source, executed.  What it cannot pin: the third-party primitives (see oracle/haiku_shim.py).

Per case the file holds: text, silence_duration, tokens, durations (seconds, as predict_duration returns them), durations after
the rules, n_frames, trailing-silence frames, the full mel of predict_mel and the trimmed mel of text2mel — float64 — plus the
float32 run's frame counts (asserted equal: cases whose float32 and float64 frame sums straddle an integer are refused).
"""
from __future__ import annotations

import hashlib
import os
import pickle
import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
OUT = Path(os.environ.get("VTTS_NAT_GOLDEN_OUT") or REPO / "tests" / "golden" / "nat_text2mel_golden.npz")  # the override: tests re-mint into a scratch file
RNG_KEY = np.array([0x1234ABCD, 0x0F1E2D3C], dtype=np.uint32)  # the synthetic checkpoint's `rng`
FRAME_SILENCE = 0.2  # --silence-duration of the whole-transcript frame-count table
CASES = [  # (transcript line index, silence_duration)
    (0, -1.0),
    (3, 0.2),
    (7, 0.05),
]


def params_digest(*dicts) -> str:
    h = hashlib.sha256()
    for d in dicts:
        for k in sorted(d):
            for n in sorted(d[k]):
                h.update(k.encode()); h.update(n.encode()); h.update(np.ascontiguousarray(d[k][n]).tobytes())
    return h.hexdigest()


def write_checkpoints(root: Path):
    """The synthetic checkpoints in the reference's on-disk format (plain dicts of numpy arrays: nothing to unpickle but numpy)."""
    sys.path.insert(0, str(REPO))
    from viettts_amd.nat.synth import synthetic_acoustic_checkpoint, synthetic_duration_checkpoint

    d = root / "assets" / "infore" / "nat"
    d.mkdir(parents=True, exist_ok=True)
    dp, ds = synthetic_duration_checkpoint()
    ap, as_ = synthetic_acoustic_checkpoint()
    with open(d / "duration_latest_ckpt.pickle", "wb") as f:
        pickle.dump({"step": 0, "params": dp, "aux": ds, "rng": RNG_KEY, "optim_state": None}, f)
    with open(d / "acoustic_latest_ckpt.pickle", "wb") as f:
        pickle.dump({"step": 0, "params": ap, "aux": as_, "rng": RNG_KEY, "optim_state": None}, f)
    sys.path.remove(str(REPO))
    return params_digest(dp, ds, ap, as_)


def main() -> int:
    if not (REF / "vietTTS/nat/model.py").exists():
        print("oracle/make_nat_golden.py: /root/reference not present — nothing minted")
        return 0
    sys.path.insert(0, str(REPO))
    from oracle import haiku_shim as shim

    sys.path.remove(str(REPO))
    shim.install()
    for m in [k for k in sys.modules if k == "vietTTS" or k.startswith("vietTTS.")]:
        del sys.modules[m]  # the repo's drop-in package of the same name must not shadow the reference
    sys.path.insert(0, str(REF))
    import vietTTS.nat.text2mel as ref_t2m  # noqa: E402  (the reference's file, unchanged)

    assert Path(ref_t2m.__file__).resolve().is_relative_to(REF), ref_t2m.__file__
    lexicon = REPO / "tests" / "golden" / "text" / "lexicon.txt"
    lines = [l.strip() for l in open(REPO / "tests" / "golden" / "text" / "transcript.txt", encoding="utf-8") if l.strip()]
    out = {"rng_key": RNG_KEY, "n_cases": np.array(len(CASES))}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        out["params_sha256"] = np.array(write_checkpoints(Path(tmp)))
        os.chdir(tmp)  # FLAGS.ckpt_dir is CWD-relative (config.py: Path("assets/infore/nat"))
        try:
            for ci, (li, sil) in enumerate(CASES):
                text = lines[li]
                res = {}
                for dt in (np.float64, np.float32):
                    shim.set_dtype(dt)
                    tokens = ref_t2m.text2tokens(text, lexicon)
                    dur = ref_t2m.predict_duration(tokens)  # [1, L] seconds
                    # text2mel.py:85-104, run as one call AND step by step (the steps' intermediates are stored)
                    mel_trim = ref_t2m.text2mel(text, lexicon, sil)
                    d2 = np.where(np.array(tokens)[None, :] == ref_t2m.FLAGS.sil_index, np.clip(dur, sil, None), dur)
                    d2 = np.where(np.array(tokens)[None, :] == ref_t2m.FLAGS.word_end_index, 0.0, d2)
                    mel_full = ref_t2m.predict_mel(tokens, d2)
                    frames = d2 * ref_t2m.FLAGS.sample_rate / (ref_t2m.FLAGS.n_fft // 4)
                    n_frames = int(np.sum(frames).item())
                    trail = int(d2[0, -1].item() * ref_t2m.FLAGS.sample_rate / (ref_t2m.FLAGS.n_fft // 4)) if tokens[-1] == ref_t2m.FLAGS.sil_index else 0
                    assert mel_full.shape == (1, n_frames, 80) and mel_trim.shape == (1, n_frames - trail, 80)
                    assert np.array_equal(mel_full[:, : n_frames - trail], mel_trim)
                    res[dt] = dict(tokens=tokens, dur=dur, d2=d2, n_frames=n_frames, trail=trail, mel_full=mel_full, mel_trim=mel_trim,
                                   frac=float(np.sum(frames)) % 1.0)
                a, b = res[np.float64], res[np.float32]
                if (a["n_frames"], a["trail"]) != (b["n_frames"], b["trail"]) or not 0.02 < a["frac"] < 0.98:
                    raise SystemExit(f"case {ci}: frame counts fp64 {a['n_frames']}/{a['trail']} vs fp32 {b['n_frames']}/{b['trail']}, "
                                     f"fractional part {a['frac']:.4f}: too close to an integer to serve as a fixture, pick another line")
                err32 = float(np.abs(a["mel_full"] - b["mel_full"]).max())
                print(f"case {ci}: line {li}, silence {sil}: {len(a['tokens'])} tokens, {a['n_frames']} frames (-{a['trail']} trailing), "
                      f"|mel| max {np.abs(a['mel_full']).max():.3f}, fp32 run vs fp64 run max-abs {err32:.2e}")
                p = f"c{ci}_"
                out[p + "text"] = np.array(text)
                out[p + "silence_duration"] = np.array(sil)
                out[p + "tokens"] = np.array(a["tokens"], dtype=np.int32)
                out[p + "durations_s"] = a["dur"].astype(np.float64)
                out[p + "durations_ruled_s"] = a["d2"].astype(np.float64)
                out[p + "n_frames"] = np.array(a["n_frames"])
                out[p + "trailing_frames"] = np.array(a["trail"])
                out[p + "mel_full"] = a["mel_full"][0].astype(np.float64)
                out[p + "mel_fp32run_maxabs"] = np.array(err32)
            # ---- integer frame counts of the WHOLE demo transcript (BASELINE: "bit-exact for the NAT duration model's integer frame counts"):
            # every line through the reference's text2tokens + predict_duration + the rules of text2mel.py:88-102, float32 as the
            # reference computes and float64 as a margin check
            allc = {k: [] for k in ("n32", "n64", "t32", "t64", "frac64", "tfrac64", "ntok")}
            for text in lines:
                per = {}
                for dt in (np.float64, np.float32):
                    shim.set_dtype(dt)
                    tokens = ref_t2m.text2tokens(text, lexicon)
                    dur = ref_t2m.predict_duration(tokens)
                    d2 = np.where(np.array(tokens)[None, :] == ref_t2m.FLAGS.sil_index, np.clip(dur, FRAME_SILENCE, None), dur)
                    d2 = np.where(np.array(tokens)[None, :] == ref_t2m.FLAGS.word_end_index, 0.0, d2).astype(dt)
                    frames = d2 * ref_t2m.FLAGS.sample_rate / (ref_t2m.FLAGS.n_fft // 4)
                    tsec = d2[0, -1].item() * ref_t2m.FLAGS.sample_rate / (ref_t2m.FLAGS.n_fft // 4) if tokens[-1] == ref_t2m.FLAGS.sil_index else 0.0
                    per[dt] = (int(np.sum(frames).item()), int(tsec), float(np.sum(frames)) % 1.0, float(tsec) % 1.0, len(tokens))
                allc["n64"].append(per[np.float64][0]); allc["t64"].append(per[np.float64][1])
                allc["n32"].append(per[np.float32][0]); allc["t32"].append(per[np.float32][1])
                allc["frac64"].append(per[np.float64][2]); allc["tfrac64"].append(per[np.float64][3]); allc["ntok"].append(per[np.float64][4])
            out["all_silence_duration"] = np.array(FRAME_SILENCE)
            out["all_n_frames_f32"] = np.array(allc["n32"]); out["all_n_frames_f64"] = np.array(allc["n64"])
            out["all_trailing_f32"] = np.array(allc["t32"]); out["all_trailing_f64"] = np.array(allc["t64"])
            out["all_frac_f64"] = np.array(allc["frac64"]); out["all_trailing_frac_f64"] = np.array(allc["tfrac64"]); out["all_n_tokens"] = np.array(allc["ntok"])
            same = int(np.sum((out["all_n_frames_f32"] == out["all_n_frames_f64"]) & (out["all_trailing_f32"] == out["all_trailing_f64"])))
            print(f"all {len(lines)} transcript lines at silence {FRAME_SILENCE}: {int(out['all_n_frames_f32'].sum())} frames; float32 and float64 runs agree on {same} lines")
        finally:
            os.chdir(cwd)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({OUT.stat().st_size / 1024:.0f} KB)")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
