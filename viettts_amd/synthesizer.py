"""Drop-in for the reference CLI ``python -m vietTTS.synthesizer`` (vietTTS/synthesizer.py:12-39).

Same flags, defaults and prints; the text normalisation applies the reference's substitutions in the
reference's order (:21-31).  Unlike the reference this module has a ``main()`` (importing the
reference runs the whole pipeline at import time).  One addition:
``--mel-file`` synthesises from a saved ``[T, 80]`` / ``[1, T, 80]`` float32 mel (.npy) so the
mel->waveform path can be driven on its own (the text path runs the NAT duration and acoustic models of
``viettts_amd/nat`` like the reference's).

    python -m viettts_amd.synthesizer --text "..." --output clip.wav --lexicon-file assets/infore/lexicon.txt
"""
from __future__ import annotations

import re
import unicodedata
from argparse import ArgumentParser
from pathlib import Path

import numpy as np

from .nat.config import FLAGS


def nat_normalize_text(text: str) -> str:
    """vietTTS/synthesizer.py:21-31 — NFKC, lower, punctuation -> " sil ", collapse."""
    sil = FLAGS.special_phonemes[FLAGS.sil_index]
    text = unicodedata.normalize("NFKC", text).lower().strip()
    steps = (
        (r"[\n.,:]+", f" {sil} "),
        ('"', " "),
        (r"\s+", " "),
        (r"[.,:;?!]+", f" {sil} "),
        ("[ ]+", " "),
        (f"( {sil}+)+ ", f" {sil} "),
    )
    for pattern, repl in steps:
        text = text.replace(pattern, repl) if pattern == '"' else re.sub(pattern, repl, text)
    return text.strip()


def build_parser() -> ArgumentParser:
    p = ArgumentParser(prog="viettts_amd.synthesizer")
    p.add_argument("--text", type=str)
    p.add_argument("--output", default="clip.wav", type=Path)
    p.add_argument("--sample-rate", default=16000, type=int)
    p.add_argument("--silence-duration", default=-1, type=float)
    p.add_argument("--lexicon-file", default=None)
    p.add_argument("--mel-file", default=None, type=Path, help="(extension) synthesise from a saved mel instead of text")
    return p


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)
    from .hifigan.mel2wave import mel2wave
    from .wavio import write_wav

    if args.mel_file is not None:
        mel = np.load(args.mel_file).astype(np.float32)
        if mel.ndim == 2:
            mel = mel[None]
    else:
        if args.text is None:
            raise SystemExit("--text (or --mel-file) is required")
        from .nat.text2mel import text2mel

        text = nat_normalize_text(args.text)
        print("Normalized text input:", text)
        mel = text2mel(text, args.lexicon_file, args.silence_duration)
    wave = mel2wave(mel)
    print("writing output to file", args.output)
    write_wav(args.output, wave, args.sample_rate)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
