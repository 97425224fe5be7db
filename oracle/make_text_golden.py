"""ORACLE tooling — pins the integer / text work of the front end to the REFERENCE's own functions.

Runs only where /root/reference exists (the build container).  The three functions that decide which token ids the
hot path's feeder sees are plain Python in the reference and need neither JAX nor Haiku once lifted out of their
modules (whose top-level imports do):

  * ``nat_normalize_text``  vietTTS/synthesizer.py:21-31   (the module runs the whole CLI at import: cannot be imported)
  * ``load_lexicon``        vietTTS/nat/text2mel.py:16-19  (module imports haiku / jax / matplotlib)
  * ``text2tokens``         vietTTS/nat/text2mel.py:37-58
  * ``load_phonemes_set``   vietTTS/nat/data_loader.py:11-13 (module imports textgrid)
  * ``FLAGS``               vietTTS/nat/config.py:8-63     (module imports jax.numpy.ndarray for type hints only)

They are lifted by AST (the function / class definitions are compiled from the reference's source text where it lies;
nothing is copied into this repo), executed on the reference's own demo workload — ``assets/transcript.txt`` with
``assets/infore/lexicon.txt`` (scripts/quick_start.sh:11-12) — and on a set of adversarial strings, and the results are
committed as ``tests/golden/text_golden.json`` together with the two data fixtures (``tests/golden/text/``), so that the
tests and ``bench.py``'s 256-sentence workload (BASELINE configs[3]) run on the GPU box without /root/reference.

Usage:  python oracle/make_text_golden.py
"""
from __future__ import annotations

import ast
import hashlib
import json
import re
import shutil
import sys
import unicodedata
from argparse import Namespace
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
OUT = REPO / "tests" / "golden"


def _lift(path: Path, names, ns):
    """exec the top-level FunctionDef / ClassDef nodes called ``names`` of ``path`` into ``ns``."""
    tree = ast.parse(path.read_text(encoding="utf-8"))
    found = []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, str(path), "exec"), ns)
            found.append(node.name)
    missing = set(names) - set(found)
    if missing:
        raise RuntimeError(f"{path}: {sorted(missing)} not found")


def reference_functions():
    ns = {"Namespace": Namespace, "Path": Path, "re": re, "unicodedata": unicodedata}
    _lift(REF / "vietTTS/nat/config.py", ["FLAGS"], ns)
    _lift(REF / "vietTTS/nat/data_loader.py", ["load_phonemes_set"], ns)
    _lift(REF / "vietTTS/nat/text2mel.py", ["load_lexicon", "text2tokens"], ns)
    _lift(REF / "vietTTS/synthesizer.py", ["nat_normalize_text"], ns)
    return ns


ADVERSARIAL = [
    "Xin chào, thế giới!",
    'He said: "hello"... and left; why?  OK!',
    "  MIXED Case\nwith\nnewlines.. and,, commas::  ",
    "sil sp spn",
    "unknownword zzzz qwerty 12345 đường",
    "ＦＵＬＬＷＩＤＴＨ ｔｅｘｔ ①②",  # NFKC changes these
    "a\tb\tc",
    "...",
    "",
    "sil sil sil , . sil",
    "tiếng việt có dấu: ngã, hỏi, nặng; sắc? huyền!",
    "abc's don't e-mail",
]


def main():
    ns = reference_functions()
    FLAGS = ns["FLAGS"]
    lex_src, tr_src = REF / "assets/infore/lexicon.txt", REF / "assets/transcript.txt"
    (OUT / "text").mkdir(parents=True, exist_ok=True)
    shutil.copyfile(lex_src, OUT / "text" / "lexicon.txt")
    shutil.copyfile(tr_src, OUT / "text" / "transcript.txt")
    lexicon = ns["load_lexicon"](str(lex_src))
    lex_digest = hashlib.sha256("\n".join(f"{k}\t{v}" for k, v in sorted(lexicon.items())).encode("utf-8")).hexdigest()
    lines = [l for l in tr_src.read_text(encoding="utf-8").split("\n") if l.strip()]
    whole = tr_src.read_text(encoding="utf-8")  # scripts/quick_start.sh:11: text=`cat assets/transcript.txt` (trailing newlines dropped by the shell)
    whole = whole.rstrip("\n")
    cases = []
    for raw in lines + [whole] + ADVERSARIAL:
        norm = ns["nat_normalize_text"](raw)
        try:
            tok = ns["text2tokens"](norm, str(lex_src))
            cases.append({"raw": raw, "normalized": norm, "tokens": [int(t) for t in tok]})
        except Exception as e:  # the reference's own error behaviour is part of the contract (a lexicon phoneme outside the set)
            cases.append({"raw": raw, "normalized": norm, "error": type(e).__name__})
    golden = {
        "source": "AST-lifted from /root/reference: vietTTS/synthesizer.py:21-31, vietTTS/nat/text2mel.py:16-19,37-58, "
                  "vietTTS/nat/data_loader.py:11-13, vietTTS/nat/config.py:8-63",
        "n_transcript_lines": len(lines),
        "phonemes": ns["load_phonemes_set"](),
        "special_phonemes": FLAGS.special_phonemes,
        "sil_index": FLAGS.sil_index,
        "word_end_index": FLAGS.word_end_index,
        "flags": {k: getattr(FLAGS, k) for k in ("vocab_size", "duration_lstm_dim", "acoustic_encoder_dim", "acoustic_decoder_dim", "postnet_dim",
                                                "mel_dim", "n_fft", "sample_rate")},
        "lexicon_entries": len(lexicon),
        "lexicon_sha256": lex_digest,
        "cases": cases,
    }
    with open(OUT / "text_golden.json", "w", encoding="utf-8") as f:
        json.dump(golden, f, ensure_ascii=False, indent=0)
    n_tok = sum(len(c.get("tokens", ())) for c in cases[: len(lines)])
    print(f"wrote {OUT / 'text_golden.json'}: {len(cases)} cases ({len(lines)} transcript lines, {n_tok} tokens), lexicon {len(lexicon)} entries")


if __name__ == "__main__":
    sys.exit(main())
