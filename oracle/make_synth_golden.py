#!/usr/bin/env python
"""ORACLE — TEST INFRASTRUCTURE ONLY.  Mints tests/golden/synthesizer_golden.npz by RUNNING THE REFERENCE'S CLI.

``python -m vietTTS.synthesizer --text ... --output ... --lexicon-file ... --silence-duration ...`` — the reference's
``vietTTS/synthesizer.py`` (module-level code: argparse, ``nat_normalize_text``, ``text2mel``, ``mel2wave``, ``sf.write``) and
everything it imports (``nat/text2mel.py``, ``nat/model.py``, ``hifigan/mel2wave.py``, ``hifigan/model.py``), from where they lie
under /root/reference, unchanged, executed by ``runpy`` with ``oracle/haiku_shim.py`` standing in for haiku / jax and a
recording stand-in for ``soundfile`` (neither is installable offline).  The scratch CWD holds what the reference reads:
``assets/hifigan/config.json`` (the repo's copy of the V1 config), ``assets/infore/hifigan/hk_hifi.pickle`` (seeded synthetic V1
weights), ``assets/infore/nat/{duration,acoustic}_latest_ckpt.pickle`` (seeded synthetic checkpoints with an rng key).

The fixture holds the arguments, the two lines the CLI prints, and the float64 waveform handed to ``sf.write`` with its sample
rate.  tests/test_gpu_longform.py runs THIS repo's ``python -m vietTTS.synthesizer`` with the same arguments on the same files
and compares the WAV it writes.  Runs only where /root/reference exists.
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import pickle
import runpy
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
OUT = Path(os.environ.get("VTTS_SYNTH_GOLDEN_OUT") or REPO / "tests" / "golden" / "synthesizer_golden.npz")
TEXT_LINE, SILENCE = 0, 0.2  # transcript line (raw, with its punctuation: the CLI normalises it), --silence-duration


def write_assets(root: Path) -> str:
    """Everything the reference CLI reads, under `root` (its CWD).  Returns the digest of the NAT checkpoints."""
    sys.path.insert(0, str(REPO))
    from oracle.make_nat_golden import write_checkpoints
    from viettts_amd.hifigan.config import V1
    from viettts_amd.hifigan.synth import synthetic_params

    sys.path.remove(str(REPO))
    digest = write_checkpoints(root)
    (root / "assets/hifigan").mkdir(parents=True, exist_ok=True)
    (root / "assets/infore/hifigan").mkdir(parents=True, exist_ok=True)
    (root / "assets/hifigan/config.json").write_text((REPO / "assets/hifigan/config.json").read_text())
    with open(root / "assets/infore/hifigan/hk_hifi.pickle", "wb") as f:
        pickle.dump({k: {n: np.asarray(a) for n, a in m.items()} for k, m in synthetic_params(V1, 4321, "scaled").items()}, f)
    return digest


def main() -> int:
    if not (REF / "vietTTS/synthesizer.py").exists():
        print("oracle/make_synth_golden.py: /root/reference not present — nothing minted")
        return 0
    sys.path.insert(0, str(REPO))
    from oracle import haiku_shim as shim

    sys.path.remove(str(REPO))
    shim.install()
    shim.set_dtype(np.float64)
    written = {}
    sf = types.ModuleType("soundfile")
    sf.write = lambda path, data, samplerate: written.update(path=str(path), data=np.asarray(data), samplerate=int(samplerate))
    sys.modules["soundfile"] = sf
    for m in [k for k in sys.modules if k == "vietTTS" or k.startswith("vietTTS.")]:
        del sys.modules[m]
    sys.path.insert(0, str(REF))

    lines = [l.strip() for l in open(REPO / "tests" / "golden" / "text" / "transcript.txt", encoding="utf-8") if l.strip()]
    text = lines[TEXT_LINE]
    lexicon = REPO / "tests" / "golden" / "text" / "lexicon.txt"
    cwd, argv = os.getcwd(), sys.argv
    with tempfile.TemporaryDirectory() as tmp:
        digest = write_assets(Path(tmp))
        os.chdir(tmp)
        sys.argv = ["synthesizer", "--text", text, "--output", "out.wav", "--lexicon-file", str(lexicon), "--silence-duration", str(SILENCE)]
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                ns = runpy.run_module("vietTTS.synthesizer", run_name="__main__")
        finally:
            os.chdir(cwd)
            sys.argv = argv
    assert Path(ns["__file__"]).resolve().is_relative_to(REF), ns["__file__"]
    out_lines = [l for l in buf.getvalue().splitlines() if l.strip()]
    wave = written["data"]
    assert wave.ndim == 1 and written["path"] == "out.wav" and written["samplerate"] == 16000 and np.isfinite(wave).all()
    print("\n".join(out_lines))
    print(f"reference CLI: {wave.shape[0]} samples = {wave.shape[0] // 256} frames, |wave| max {np.abs(wave).max():.3f}")
    np.savez_compressed(OUT, text=np.array(text), lexicon=np.array("tests/golden/text/lexicon.txt"), silence_duration=np.array(SILENCE),
                        stdout=np.array(json.dumps(out_lines)), wave=wave.astype(np.float64), samplerate=np.array(written["samplerate"]),
                        nat_params_sha256=np.array(digest))
    print(f"wrote {OUT} ({OUT.stat().st_size / 1024:.0f} KB)")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
