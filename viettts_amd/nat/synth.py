"""Deterministic synthetic checkpoints of the NAT duration model (no pretrained NAT checkpoint exists offline: the
reference fetches them with wget, scripts/quick_start.sh).  Haiku-layout ``params`` / ``state`` dicts with the module
paths text2mel.py:23-24 produces (``duration_model/~/...``), drawn from a CPU numpy Generator so that a seed gives
the same bits on every box.  Scales keep the activations O(1) through the stack (fan-in scaled weights, BatchNorm
statistics near (0, 1)) and the predicted durations around 0.1 s per token."""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from .config import FLAGS

HaikuDict = Dict[str, Dict[str, np.ndarray]]


def synthetic_duration_checkpoint(seed: int = 777, vocab_size: int = FLAGS.vocab_size, dim: int = FLAGS.duration_lstm_dim) -> Tuple[HaikuDict, HaikuDict]:
    g = np.random.default_rng(seed)
    f32 = np.float32

    def n(*shape, scale=1.0):
        return (g.standard_normal(shape) * scale).astype(f32)

    pre = "duration_model/~/token_encoder/~/"
    P: HaikuDict = {pre + "embed": {"embeddings": n(vocab_size, dim)}}
    S: HaikuDict = {}
    for i in range(3):
        sfx = f"_{i}" if i else ""
        P[pre + "conv1_d" + sfx] = {"w": n(3, dim, dim, scale=(2.0 / (3 * dim)) ** 0.5), "b": n(dim, scale=0.05)}
        P[pre + "batch_norm" + sfx] = {"scale": (1.0 + n(1, 1, dim, scale=0.1)).astype(f32), "offset": n(1, 1, dim, scale=0.1)}
        S[pre + "batch_norm" + sfx + "/~/mean_ema"] = {"average": n(1, 1, dim, scale=0.2), "hidden": n(1, 1, dim), "counter": np.array(1000, np.int32)}
        S[pre + "batch_norm" + sfx + "/~/var_ema"] = {"average": (1.0 + np.abs(n(1, 1, dim, scale=0.3))).astype(f32), "hidden": n(1, 1, dim), "counter": np.array(1000, np.int32)}
    for l in ("lstm", "lstm_1"):
        P[pre + l + "/linear"] = {"w": n(2 * dim, 4 * dim, scale=(1.0 / (2 * dim)) ** 0.5), "b": n(4 * dim, scale=0.05)}
    P["duration_model/~/linear"] = {"w": n(2 * dim, dim, scale=(1.0 / (2 * dim)) ** 0.5 * 3.0), "b": n(dim, scale=0.05)}
    P["duration_model/~/linear_1"] = {"w": n(dim, 1, scale=(1.0 / dim) ** 0.5 * 2.0), "b": np.array([-2.0], f32)}
    return P, S


def synthetic_acoustic_checkpoint(seed: int = 778, vocab_size: int = FLAGS.vocab_size, enc: int = FLAGS.acoustic_encoder_dim,
                                  dec: int = FLAGS.acoustic_decoder_dim, prenet: int = 256, mel: int = FLAGS.mel_dim,
                                  post: int = FLAGS.postnet_dim) -> Tuple[HaikuDict, HaikuDict]:
    """Haiku-layout ``params`` / ``state`` of AcousticModel (model.py:76-93) with seeded values; scales keep the
    autoregressive loop stable (mel outputs O(1), LSTM pre-activations O(1))."""
    g = np.random.default_rng(seed)
    f32 = np.float32

    def n(*shape, scale=1.0):
        return (g.standard_normal(shape) * scale).astype(f32)

    P: HaikuDict = {}
    S: HaikuDict = {}

    def bn(path, C):
        P[path] = {"scale": (1.0 + n(1, 1, C, scale=0.1)).astype(f32), "offset": n(1, 1, C, scale=0.1)}
        S[path + "/~/mean_ema"] = {"average": n(1, 1, C, scale=0.2), "hidden": n(1, 1, C), "counter": np.array(1000, np.int32)}
        S[path + "/~/var_ema"] = {"average": (1.0 + np.abs(n(1, 1, C, scale=0.3))).astype(f32), "hidden": n(1, 1, C), "counter": np.array(1000, np.int32)}

    te = "acoustic_model/~/token_encoder/~/"
    P[te + "embed"] = {"embeddings": n(vocab_size, enc)}
    for i in range(3):
        sfx = f"_{i}" if i else ""
        P[te + "conv1_d" + sfx] = {"w": n(3, enc, enc, scale=(2.0 / (3 * enc)) ** 0.5), "b": n(enc, scale=0.05)}
        bn(te + "batch_norm" + sfx, enc)
    for l in ("lstm", "lstm_1"):
        P[te + l + "/linear"] = {"w": n(2 * enc, 4 * enc, scale=(1.0 / (2 * enc)) ** 0.5), "b": n(4 * enc, scale=0.05)}
    X = 2 * enc + prenet
    pre = "acoustic_model/~/"
    P[pre + "lstm/linear"] = {"w": n(X + dec, 4 * dec, scale=(1.0 / (X + dec)) ** 0.5), "b": n(4 * dec, scale=0.05)}
    P[pre + "lstm_1/linear"] = {"w": n(X + dec + dec, 4 * dec, scale=(1.0 / (2 * dec + X)) ** 0.5), "b": n(4 * dec, scale=0.05)}
    P[pre + "linear"] = {"w": n(2 * dec, mel, scale=(1.0 / (2 * dec)) ** 0.5 * 2.0), "b": n(mel, scale=0.1)}
    P[pre + "linear_1"] = {"w": n(mel, prenet, scale=(2.0 / mel) ** 0.5)}
    P[pre + "linear_2"] = {"w": n(prenet, prenet, scale=(2.0 / prenet) ** 0.5)}
    for i in range(5):
        sfx = f"_{i}" if i else ""
        cin, cout = (mel if i == 0 else post), (mel if i == 4 else post)
        P[pre + "conv1_d" + sfx] = {"w": n(5, cin, cout, scale=(1.0 / (5 * cin)) ** 0.5), "b": n(cout, scale=0.05)}
        if i < 4:
            bn(pre + "batch_norm" + sfx, post)
    return P, S


def synthetic_sentences(n: int = 256, seed: int = 2024):
    """Token-id lists shaped like lexicon output (sil, words of 2-5 phonemes each closed by the word-end token, sil):
    6-21 words per sentence.  BASELINE.json configs[3] names 256 InfoRe sentences; transcripts and lexicon do not
    travel to the GPU box, the token statistics do."""
    from .config import FLAGS

    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        body = []
        for _ in range(int(rng.integers(6, 22))):
            body += [int(v) for v in rng.integers(4, 90, size=int(rng.integers(2, 6)))] + [FLAGS.word_end_index]
        out.append([FLAGS.sil_index] + body + [FLAGS.sil_index])
    return out


def transcript_sentences(n: int, transcript_fn, lexicon_fn):
    """BASELINE.json configs[3]'s workload: the first ``n`` non-empty lines CYCLED from the reference's demo
    transcript (assets/transcript.txt, 26 lines; scripts/quick_start.sh:11-12), each normalised by the CLI's rules
    (synthesizer.py:21-31) and tokenised against the InfoRe lexicon (text2mel.py:37-58).  Returns token-id lists."""
    from ..synthesizer import nat_normalize_text
    from .text2mel import text2tokens

    with open(transcript_fn, "r", encoding="utf-8") as f:
        lines = [l for l in f.read().split("\n") if l.strip()]
    if not lines:
        raise ValueError(f"{transcript_fn}: no sentences")
    toks = [text2tokens(nat_normalize_text(l), lexicon_fn) for l in lines]
    return [list(toks[i % len(toks)]) for i in range(n)]
