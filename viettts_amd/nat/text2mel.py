"""Call surface of ``vietTTS.nat.text2mel`` (vietTTS/nat/text2mel.py).

What is built (SURVEY.md §8b "companion entry", §8f rank 2 groundwork):
  * ``load_lexicon`` / ``text2tokens`` — the deterministic text -> token-id front end (:16-19, :37-58);
  * the two INTEGER quantities BASELINE.json wants bit-exact — ``n_frames`` (:78-79) and the trailing
    ``silence_frame`` (:99-101) — as pure functions of the fp32 duration vector, computed with the
    same dtype and operations (fp32 multiply by 16000, fp32 divide by 256, fp32 sum, truncate; the order of the sum's
    additions is numpy's, see :func:`n_frames_from_durations`);
  * ``text2mel(text, lexicon_fn, silence_duration)`` with the reference's signature.

  * ``predict_duration(tokens)`` (:22-34) on the MI355X: the NAT duration model (vietTTS/nat/model.py:9-70) runs in
    the HIP library behind include/vtts_nat.h (viettts_amd/nat/duration.py); there is no CPU path.

  * ``predict_mel(tokens, durations)`` (:61-82) on the MI355X: the acoustic network's inference path
    (vietTTS/nat/model.py:128-151) behind the same header (viettts_amd/nat/acoustic.py).  The reference's prenet
    dropout is ON at inference and draws from JAX's threefry PRNG through Haiku's key chain, starting from the
    checkpoint's ``rng``: with a checkpoint loaded ``predict_mel`` draws THAT stream on the GPU (restated in round 2 for
    jax.random's classic layout, pinned by JAX's documented known answers, not by a JAX run); without a key it draws this
    library's own per-sentence stream from ``dropout_seed`` (Threefry-2x32-20, include/vtts_nat.h).

``text2mel`` uses a registered mel provider (:func:`set_mel_provider`; tests and the CLI's ``--mel-file``) if there is
one, else the two networks, loading ``duration_latest_ckpt.pickle`` / ``acoustic_latest_ckpt.pickle`` from
``FLAGS.ckpt_dir`` on first use (``FileNotFoundError`` if absent, as in the reference).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from .config import FLAGS, load_phonemes_set

FRAMES_PER_SECOND_NUM = FLAGS.sample_rate  # durations * sample_rate / (n_fft // 4), text2mel.py:78
FRAMES_PER_SECOND_DEN = FLAGS.n_fft // 4


def load_lexicon(fn) -> Dict[str, str]:
    """word<TAB>phonemes per line, lower-cased and stripped (text2mel.py:16-19).  As the reference's ``dict(lines)``,
    a line that does not split into exactly two tab-separated fields raises ``ValueError``; a repeated word keeps its
    LAST entry."""
    with open(fn, "r", encoding="utf-8") as f:
        lines = [l.lower().strip().split("\t") for l in f.readlines()]
    for n, parts in enumerate(lines):
        if len(parts) != 2:
            raise ValueError(f"dictionary update sequence element #{n} has length {len(parts)}; 2 is required")
    return dict(lines)


def text2tokens(text: str, lexicon_fn) -> List[int]:
    """sil + per word (special phoneme | lexicon phonemes + word_end | known letters + word_end) + sil
    (text2mel.py:37-58).  Token ids are bit-exact with the reference's own function on its demo workload and on
    adversarial strings (tests/golden/text_golden.json, minted by oracle/make_text_golden.py); a lexicon entry that
    names a phoneme outside the set raises ``ValueError`` as the reference's ``phonemes.index`` does."""
    phonemes = load_phonemes_set()
    index = {p: i for i, p in enumerate(phonemes)}
    lexicon = load_lexicon(lexicon_fn)
    tokens = [FLAGS.sil_index]
    for word in text.strip().lower().split():
        if word in FLAGS.special_phonemes:
            tokens.append(index[word])
        elif word in lexicon:
            for p in lexicon[word].split():
                if p not in index:
                    raise ValueError(f"{p!r} is not in list")
                tokens.append(index[p])
            tokens.append(FLAGS.word_end_index)
        else:
            tokens.extend(index[ch] for ch in word if ch in index)
            tokens.append(FLAGS.word_end_index)
    tokens.append(FLAGS.sil_index)
    return tokens


def apply_duration_rules(tokens: Sequence[int], durations: np.ndarray, silence_duration: float) -> np.ndarray:
    """sil tokens: clip(duration, min=silence_duration); word-end tokens: 0 (text2mel.py:90-97).
    ``durations`` is float32 ``[1, L]`` in seconds."""
    d = np.asarray(durations, dtype=np.float32).copy()
    tok = np.asarray(tokens)[None, :]
    d = np.where(tok == FLAGS.sil_index, np.maximum(d, np.float32(silence_duration)), d).astype(np.float32)
    d = np.where(tok == FLAGS.word_end_index, np.float32(0.0), d).astype(np.float32)
    return d


def durations_to_frames(durations: np.ndarray) -> np.ndarray:
    """seconds -> frames in fp32 with the reference's operation order:
    ``durations * sample_rate / (n_fft // 4)`` (text2mel.py:78) = (d * 16000f) / 256f."""
    d = np.asarray(durations, dtype=np.float32)
    return (d * np.float32(FRAMES_PER_SECOND_NUM)) / np.float32(FRAMES_PER_SECOND_DEN)


def n_frames_from_durations(durations: np.ndarray) -> int:
    """``int(jnp.sum(durations_in_frames).item())`` (text2mel.py:79): fp32 sum, truncation.

    Exact in dtype, operand values and truncation.  The ORDER of the fp32 additions is numpy's (pairwise) here and XLA's in the
    reference (unspecified, backend- and version-dependent): a sum that lands within an fp32 ulp of an integer can truncate
    to a neighbouring frame count (one frame = 256 samples).  GPU vs oracle agree bit for bit (tests/test_gpu_nat.py: both use
    this function on bit-identical durations); agreement with a JAX run is unverified (no jax offline)."""
    return int(np.sum(durations_to_frames(durations), dtype=np.float32))


def trailing_silence_frames(durations: np.ndarray) -> int:
    """``int(end_silence * sample_rate / (n_fft // 4))`` with ``end_silence`` a Python float
    (text2mel.py:99-101): the fp32 duration is widened to double by ``.item()`` first."""
    end_silence = float(np.asarray(durations, dtype=np.float32)[0, -1])
    return int(end_silence * FLAGS.sample_rate / (FLAGS.n_fft // 4))


def frame_plan(token_lists, seconds, silence_duration: float):
    """The frame rules of text2mel.py:78-79, :90-102 for MANY sentences at once (viettts_amd/pipeline.py): per sentence the
    per-token durations in frames (float32), ``n_frames`` and the trailing-silence frame count — bit for bit what
    :func:`apply_duration_rules` / :func:`durations_to_frames` / :func:`n_frames_from_durations` /
    :func:`trailing_silence_frames` give sentence by sentence (tests/test_frontend_cpu.py).  The elementwise rules run once on a
    padded ``[B, Lmax]`` array; the float32 sum of a sentence is taken over ITS tokens only (the additions' order depends on
    the count)."""
    B = len(token_lists)
    lens = [len(t) for t in token_lists]
    Lmax = max(lens) if B else 0
    tok = np.full((B, Lmax), -1, dtype=np.int64)
    d = np.zeros((B, Lmax), dtype=np.float32)
    for i, (t, s) in enumerate(zip(token_lists, seconds)):
        tok[i, : lens[i]] = t
        d[i, : lens[i]] = np.asarray(s, dtype=np.float32).reshape(-1)
    d = np.where(tok == FLAGS.sil_index, np.maximum(d, np.float32(silence_duration)), d).astype(np.float32)
    d = np.where(tok == FLAGS.word_end_index, np.float32(0.0), d).astype(np.float32)
    fr = (d * np.float32(FRAMES_PER_SECOND_NUM)) / np.float32(FRAMES_PER_SECOND_DEN)
    frames, nfr, trail = [], [], []
    for i in range(B):
        row = fr[i, : lens[i]]
        frames.append(row)
        nfr.append(int(np.sum(row[None, :], dtype=np.float32)))
        if lens[i] and tok[i, lens[i] - 1] == FLAGS.sil_index:
            trail.append(int(float(d[i, lens[i] - 1]) * FLAGS.sample_rate / (FLAGS.n_fft // 4)))
        else:
            trail.append(0)
    return frames, nfr, trail


_DURATION_MODEL = None


def load_duration_checkpoint(path=None):
    """``dic["params"], dic["aux"]`` of ``duration_latest_ckpt.pickle`` (text2mel.py:27-28, written by
    vietTTS/nat/utils.py:18-24) as ``{module: {name: ndarray}}``.  Checkpoints pickled as plain numpy dicts load as they are; ones
    holding Haiku mappings / jax arrays are read without those libraries, best effort (viettts_amd/nat/ckpt.py)."""
    from .ckpt import load_checkpoint

    path = FLAGS.ckpt_dir / "duration_latest_ckpt.pickle" if path is None else path
    dic = load_checkpoint(path)
    return dic["params"], dic["aux"]


def set_duration_model(model) -> None:
    """Install a loaded :class:`viettts_amd.nat.duration.DurationModel` for :func:`predict_duration` (tests install
    one with synthetic weights; by default the checkpoint under FLAGS.ckpt_dir is loaded on first use)."""
    global _DURATION_MODEL
    _DURATION_MODEL = model


def predict_duration(tokens: Sequence[int]) -> np.ndarray:
    """Reference signature and result (text2mel.py:22-34): float32 ``[1, L]`` seconds per token, computed on the GPU."""
    global _DURATION_MODEL
    if _DURATION_MODEL is None:
        from .duration import DurationModel

        params, state = load_duration_checkpoint()  # FileNotFoundError if absent, as in the reference
        m = DurationModel()
        m.load_params(params, state)
        _DURATION_MODEL = m
    return _DURATION_MODEL([list(tokens)])[0][None, :]


_ACOUSTIC_MODEL = None


def set_acoustic_model(model) -> None:
    """Install a loaded :class:`viettts_amd.nat.acoustic.AcousticModel` for :func:`predict_mel`."""
    global _ACOUSTIC_MODEL
    _ACOUSTIC_MODEL = model


def load_acoustic_checkpoint(path=None, with_rng: bool = False):
    """``dic["params"], dic["aux"]`` (and, ``with_rng``, ``dic["rng"]`` as uint32[2]) of ``acoustic_latest_ckpt.pickle``
    (text2mel.py:62-71); read as :func:`load_duration_checkpoint` reads its file."""
    from .ckpt import load_checkpoint

    path = FLAGS.ckpt_dir / "acoustic_latest_ckpt.pickle" if path is None else path
    dic = load_checkpoint(path)
    if with_rng:
        rng = dic.get("rng")
        return dic["params"], dic["aux"], (None if rng is None else np.asarray(rng).astype(np.uint32).reshape(-1)[-2:])
    return dic["params"], dic["aux"]


def predict_mel(tokens: Sequence[int], durations: np.ndarray, dropout_seed: Optional[int] = 0, dropout_rng=None) -> np.ndarray:
    """Reference signature and result (text2mel.py:61-82): ``durations`` float32 ``[1, L]`` in SECONDS -> mel
    ``[1, n_frames, 80]`` with ``n_frames = int(sum(durations * sample_rate / hop))``.

    Prenet dropout (always on in the reference, model.py:95-100): with ``dropout_rng`` (a jax PRNGKey, uint32[2]) — or, by
    default, the ``rng`` of the checkpoint this function loaded, as text2mel.py:65-73 passes it to ``forward.apply`` — the
    masks are the reference's own stream (jax.random's classic threefry layout under Haiku's key chain, drawn on the GPU);
    without a key, ``dropout_seed`` seeds this library's own stream; ``dropout_seed=None`` and no key: no dropout."""
    global _ACOUSTIC_MODEL
    if _ACOUSTIC_MODEL is None:
        from .acoustic import AcousticModel

        params, state, rng = load_acoustic_checkpoint(with_rng=True)
        m = AcousticModel()
        m.load_params(params, state)
        m.checkpoint_rng = rng
        _ACOUSTIC_MODEL = m
    if dropout_rng is None:
        dropout_rng = getattr(_ACOUSTIC_MODEL, "checkpoint_rng", None)
    frames = durations_to_frames(durations)  # :78
    n_frames = n_frames_from_durations(durations)  # :79
    if n_frames < 1:
        return np.zeros((1, 0, FLAGS.mel_dim), dtype=np.float32)
    if dropout_rng is not None:  # the reference's own stream
        return _ACOUSTIC_MODEL([list(tokens)], [frames[0]], [n_frames], dropout_rng=dropout_rng)[0][None]
    seeds = None if dropout_seed is None else [dropout_seed]  # masks drawn on the GPU (include/vtts_nat.h)
    return _ACOUSTIC_MODEL([list(tokens)], [frames[0]], [n_frames], dropout_seeds=seeds)[0][None]


_MEL_PROVIDER: Optional[Callable] = None


def set_mel_provider(fn: Optional[Callable]) -> None:
    """Register ``fn(tokens, lexicon_fn, silence_duration) -> mel [1, T, 80]`` standing in for the NAT
    networks until those rows are built."""
    global _MEL_PROVIDER
    _MEL_PROVIDER = fn


def text2mel(text: str, lexicon_fn=FLAGS.data_dir / "lexicon.txt", silence_duration: float = -1.0):
    """Reference signature (text2mel.py:85-87).  Returns ``[1, T, 80]`` float32 log-mel."""
    tokens = text2tokens(text, lexicon_fn)
    if _MEL_PROVIDER is not None:
        mel = np.asarray(_MEL_PROVIDER(tokens, lexicon_fn, silence_duration), dtype=np.float32)
        if mel.ndim != 3 or mel.shape[0] != 1 or mel.shape[2] != FLAGS.mel_dim:
            raise ValueError(f"mel provider returned shape {mel.shape}, expected [1, T, {FLAGS.mel_dim}]")
        return mel
    durations = predict_duration(tokens)  # :89
    durations = apply_duration_rules(tokens, durations, silence_duration)  # :90-97
    mels = predict_mel(tokens, durations)  # :98
    if tokens[-1] == FLAGS.sil_index:  # :99-102
        silence_frame = trailing_silence_frames(durations)
        mels = mels[:, : (mels.shape[1] - silence_frame)]
    return mels
