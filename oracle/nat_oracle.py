"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU (numpy) restatement of the INFERENCE side of the reference's NAT networks
(vietTTS/nat/model.py): ``DurationModel`` (:53-70) and ``AcousticModel.inference`` (:128-151), both built on
``TokenEncoder`` (:9-50), taking Haiku-layout parameter / state dicts.

**Parity: the WIRING is pinned to the reference's own code, executed; the third-party primitives are not.**
* Pinned (round 2): ``oracle/make_nat_golden.py`` imports the reference's ``vietTTS/nat/text2mel.py`` and ``vietTTS/nat/model.py``
  from /root/reference and runs ``text2mel`` / ``predict_duration`` / ``predict_mel`` UNCHANGED on seeded synthetic checkpoints,
  with ``oracle/haiku_shim.py`` standing in for haiku / jax; the fixture ``tests/golden/nat_text2mel_golden.npz`` holds what they
  return.  ``duration_model`` below reproduces it bit for bit and ``acoustic_inference`` (with ``haiku_prenet_keep_masks``) to
  3e-15 (tests/test_nat_cpu.py): layer order, masks, the skip-connection order, the order of rng draws behind the always-on
  prenet dropout, text2mel's silence rules and frame arithmetic are the reference's.
* **Primitives: arithmetic pinned against PyTorch's independent implementations (round 5), conventions by reading.**
  ``tests/test_nat_primitives_torch_cpu.py`` compares ``lstm_step`` (one step and a whole sequence) with ``torch.nn.LSTMCell`` / ``nn.LSTM``
  through the written-out gate permutation, ``batchnorm_eval`` with ``BatchNorm1d.eval()``, ``gelu_tanh`` with ``F.gelu(approximate="tanh")``,
  ``softplus``, ``conv1d_same`` with ``F.conv1d`` (pads (k-1)//2, k//2; odd and even k) and ``gaussian_upsample`` with a torch softmax / einsum,
  all to 1e-12 in float64 — so the arithmetic of each primitive is no longer a single reading shared with ``oracle/haiku_shim.py``.
  What stays **unpinned by a JAX run** are Haiku's / JAX's CONVENTIONS listed below (gate order i, g, f, o and the +1 forget bias, weight layouts,
  gelu's default form, SAME's pad split).  The reference's tests for these modules (tests/test_nat_duration.py,
  tests/test_nat_acoustic.py) assert output SHAPES only, ship no vector, and need jax + dm-haiku, which cannot be installed here
  (SURVEY.md §4, Appendix D); no NAT checkpoint ships with the reference either.  The shim's primitives ARE the functions below, so
  a wrong reading of one of them is shared and not caught.  What is restated is the published behaviour of the third-party modules
  the reference calls (dm-haiku / jax, unpinned in setup.py:6-19):

* ``hk.Embed``        gather rows of ``embeddings[V, D]``
* ``hk.Conv1D(C, k)`` default ``padding="SAME"``, stride 1: cross-correlation, ``w[k, Cin, Cout]``, pads ((k-1)//2, k//2)
* ``hk.BatchNorm(True, True, 0.9)`` with ``is_training=False``:
  ``(x - mean_ema) * scale * rsqrt(var_ema + 1e-5) + offset`` with the EMA ``average`` values of the state dict
* ``hk.LSTM(H)``      ``gates = concat[x, h] @ w + b``; split order i, g, f, o; ``f = sigmoid(f + 1)``;
  ``c' = f*c + sigmoid(i)*tanh(g)``; ``h' = sigmoid(o)*tanh(c')``
* ``hk.ResetCore``    state <- initial state where ``should_reset``; at inference the mask (model.py:38) is True
  only from position ``lengths - 1`` on, i.e. after flipping only at the first backward steps, where the state IS
  the initial state (a no-op for ``lengths == L``; kept general here)
* ``hk.deep_rnn_with_skip_connections([LSTM, LSTM])`` (dm-haiku ``recurrent.py``, ``_DeepRNN.__call__``): for layer
  idx > 0 ``current_inputs = tree_map(concat, inputs, current_inputs)``, i.e. layer 2 sees
  ``concat[network input, layer-1 output]`` (the INPUT first); ``hk.LSTM`` then appends its own hidden state, so
  layer 2's weight rows are ``[x ; h1 ; h2]``; the network output is ``concat`` of both layers' outputs
* ``jax.nn.gelu`` default ``approximate=True`` (tanh form), ``jax.nn.softplus = logaddexp(x, 0)``
* ``hk.dropout(key, rate, x)``: ``keep = bernoulli(key, 1 - rate)``; ``where(keep, x / (1 - rate), 0)``.  The keys
  come from JAX's threefry PRNG through Haiku's key chain; callers pass the keep masks explicitly (``prenet_masks``), so
  two implementations can be compared on identical masks, and ``haiku_prenet_keep_masks`` (end of this file) restates the
  reference's own stream for jax.random's classic layout.

Only ``tests/`` may import this module.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import numpy as np

BN_EPS = 1e-5  # hk.BatchNorm default eps


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def gelu_tanh(x):
    """jax.nn.gelu(approximate=True)."""
    c = x.dtype.type(np.sqrt(2.0 / np.pi))
    return x.dtype.type(0.5) * x * (x.dtype.type(1.0) + np.tanh(c * (x + x.dtype.type(0.044715) * x * x * x)))


def softplus(x):
    """jax.nn.softplus = logaddexp(x, 0)."""
    return np.logaddexp(x, x.dtype.type(0.0))


def conv1d_same(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """hk.Conv1D(..., padding="SAME") on ``x [L, Cin]`` with ``w [k, Cin, Cout]`` (model.py:16-18, :91-92)."""
    k = w.shape[0]
    pl, pr = (k - 1) // 2, k // 2
    L = x.shape[0]
    xp = np.zeros((L + pl + pr, x.shape[1]), dtype=x.dtype)
    xp[pl : pl + L] = x
    y = np.broadcast_to(b.astype(x.dtype), (L, w.shape[2])).copy()
    for j in range(k):
        y += xp[j : j + L] @ w[j].astype(x.dtype)
    return y


def batchnorm_eval(x, scale, offset, mean, var):
    """hk.BatchNorm(create_scale, create_offset, decay) with is_training=False (model.py:19-21, :29-36)."""
    dt = x.dtype
    inv = scale.reshape(-1).astype(dt) / np.sqrt(var.reshape(-1).astype(dt) + dt.type(BN_EPS))
    return (x - mean.reshape(-1).astype(dt)) * inv + offset.reshape(-1).astype(dt)


def lstm_step(x, h, c, w, b):
    """hk.LSTM.__call__: gate order i, g, f, o; +1 on the forget gate."""
    H = h.shape[-1]
    gates = np.concatenate([x, h], axis=-1) @ w + b
    i, g, f, o = gates[..., :H], gates[..., H : 2 * H], gates[..., 2 * H : 3 * H], gates[..., 3 * H :]
    one = gates.dtype.type(1.0)
    c2 = sigmoid(f + one) * c + sigmoid(i) * np.tanh(g)
    h2 = sigmoid(o) * np.tanh(c2)
    return h2, c2


class Params:
    """Access to a Haiku ``params`` / ``state`` pair by module-path suffix (the top-level prefix depends on how the
    reference wrapped the module in hk.transform; only the tail is architectural)."""

    def __init__(self, params: Dict[str, Dict[str, np.ndarray]], state: Dict[str, Dict[str, np.ndarray]], dtype):
        self.p, self.s, self.dt = params, state, dtype

    def get(self, suffix: str, name: str, state: bool = False) -> np.ndarray:
        src = self.s if state else self.p
        hits = [k for k in src if k == suffix or k.endswith("/" + suffix)]
        if len(hits) != 1:
            raise KeyError(f"{suffix!r}: {len(hits)} matches among {sorted(src)[:6]}...")
        return np.asarray(src[hits[0]][name]).astype(self.dt)


def token_encoder(P: Params, prefix: str, tokens: np.ndarray, length: int) -> np.ndarray:
    """TokenEncoder.__call__ with is_training=False (model.py:26-50) for ONE sequence ``tokens [L]``.
    Returns ``[L, 2*H]`` = concat(forward LSTM outputs, time-flipped backward LSTM outputs)."""
    dt = P.dt
    x = P.get(f"{prefix}/~/embed", "embeddings")[tokens]  # :27
    for i, (cv, bn) in enumerate((("conv1_d", "batch_norm"), ("conv1_d_1", "batch_norm_1"), ("conv1_d_2", "batch_norm_2"))):
        x = conv1d_same(x, P.get(f"{prefix}/~/{cv}", "w"), P.get(f"{prefix}/~/{cv}", "b"))
        x = batchnorm_eval(
            x,
            P.get(f"{prefix}/~/{bn}", "scale"),
            P.get(f"{prefix}/~/{bn}", "offset"),
            P.get(f"{prefix}/~/{bn}/~/mean_ema", "average", state=True),
            P.get(f"{prefix}/~/{bn}/~/var_ema", "average", state=True),
        )
        x = np.maximum(x, dt(0))  # jax.nn.relu (:28, :31, :34)
    L = x.shape[0]
    H = P.get(f"{prefix}/~/lstm/linear", "b").shape[0] // 4
    mask = np.arange(L) >= (length - 1)  # :38
    wf, bf_ = P.get(f"{prefix}/~/lstm/linear", "w"), P.get(f"{prefix}/~/lstm/linear", "b")
    wb, bb = P.get(f"{prefix}/~/lstm_1/linear", "w"), P.get(f"{prefix}/~/lstm_1/linear", "b")
    h = np.zeros(H, dt)
    c = np.zeros(H, dt)
    fwd = np.empty((L, H), dt)
    for t in range(L):  # :39-40
        h, c = lstm_step(x[t], h, c, wf, bf_)
        fwd[t] = h
    h = np.zeros(H, dt)
    c = np.zeros(H, dt)
    bwd = np.empty((L, H), dt)
    xb, mb = x[::-1], mask[::-1]  # :41
    for t in range(L):  # :42-45 — hk.ResetCore: reset BEFORE the step where the flag is set
        if mb[t]:
            h = np.zeros(H, dt)
            c = np.zeros(H, dt)
        h, c = lstm_step(xb[t], h, c, wb, bb)
        bwd[t] = h
    return np.concatenate([fwd, bwd[::-1]], axis=-1)  # :46


def duration_model(params, state, tokens: np.ndarray, length: Optional[int] = None, dtype=np.float32) -> np.ndarray:
    """DurationModel(is_training=False)(DurationInput(tokens[None], [len], None))[0] (model.py:53-70,
    text2mel.py:22-34): seconds per token, ``[L]``."""
    P = Params(params, state, dtype)
    tokens = np.asarray(tokens, dtype=np.int64)
    L = tokens.shape[0]
    x = token_encoder(P, "duration_model/~/token_encoder", tokens, L if length is None else length)
    x = x @ P.get("duration_model/~/linear", "w") + P.get("duration_model/~/linear", "b")  # :64-66
    x = gelu_tanh(x)
    x = x @ P.get("duration_model/~/linear_1", "w") + P.get("duration_model/~/linear_1", "b")
    return softplus(x[:, 0])  # :69-70


def gaussian_upsample(x: np.ndarray, durations: np.ndarray, n_frames: int) -> np.ndarray:
    """AcousticModel.upsample (model.py:102-111) for one sequence: ``x [T, D]``, ``durations [T]`` in FRAMES."""
    dt = x.dtype
    ruler = np.arange(n_frames, dtype=dt)
    end_pos = np.cumsum(durations.astype(dt))
    mid_pos = end_pos - durations.astype(dt) / dt.type(2)
    d2 = np.square(mid_pos[None, :] - ruler[:, None]) / dt.type(10.0)
    z = -d2
    z = z - z.max(axis=-1, keepdims=True)
    w = np.exp(z)
    w = w / w.sum(axis=-1, keepdims=True)
    return w @ x


def acoustic_inference(
    params,
    state,
    tokens: np.ndarray,
    durations_frames: np.ndarray,
    n_frames: int,
    prenet_masks: Optional[Callable[[int], Tuple[np.ndarray, np.ndarray]]] = None,
    dtype=np.float32,
) -> np.ndarray:
    """AcousticModel(is_training=False).inference(tokens[None], durations[None], n_frames)[0] (model.py:128-151).

    ``prenet_masks(t)`` returns the two boolean KEEP masks ``[256]`` of frame t's prenet dropout (rate 0.5, always on:
    model.py:95-100); ``None`` = no dropout (the expectation-free deterministic variant used to compare
    implementations).  Returns the mel ``[n_frames, mel_dim]``."""
    P = Params(params, state, dtype)
    dt = dtype
    tokens = np.asarray(tokens, dtype=np.int64)
    pre = "acoustic_model"
    x = token_encoder(P, f"{pre}/~/token_encoder", tokens, tokens.shape[0])  # :131
    cond = gaussian_upsample(x, np.asarray(durations_frames, dtype=dt), n_frames)  # :132
    w1, b1 = P.get(f"{pre}/~/lstm/linear", "w"), P.get(f"{pre}/~/lstm/linear", "b")
    w2, b2 = P.get(f"{pre}/~/lstm_1/linear", "w"), P.get(f"{pre}/~/lstm_1/linear", "b")
    H = b1.shape[0] // 4
    wp, bp = P.get(f"{pre}/~/linear", "w"), P.get(f"{pre}/~/linear", "b")  # projection (:85)
    f1, f2 = P.get(f"{pre}/~/linear_1", "w"), P.get(f"{pre}/~/linear_2", "w")  # prenet_fc1/2, no bias (:88-89)
    mel_dim = wp.shape[1]
    prev = np.zeros(mel_dim, dt)
    h1 = np.zeros(H, dt); c1 = np.zeros(H, dt); h2 = np.zeros(H, dt); c2 = np.zeros(H, dt)
    out = np.empty((n_frames, mel_dim), dt)
    two = dt(2.0)
    for t in range(n_frames):  # loop_fn (:134-141)
        p = np.maximum(prev @ f1, dt(0))
        if prenet_masks is not None:
            k1, k2 = prenet_masks(t)
            p = np.where(k1, p * two, dt(0))
        p = np.maximum(p @ f2, dt(0))
        if prenet_masks is not None:
            p = np.where(k2, p * two, dt(0))
        xin = np.concatenate([cond[t], p])
        h1, c1 = lstm_step(xin, h1, c1, w1, b1)
        h2, c2 = lstm_step(np.concatenate([xin, h1]), h2, c2, w2, b2)  # skip connection: input first
        prev = np.concatenate([h1, h2]) @ wp + bp
        out[t] = prev
    # postnet (:113-121): 4 x (Conv1D(512, 5) + BatchNorm + tanh) + Conv1D(mel_dim, 5); residual added (:151)
    y = out
    for i in range(5):
        cv = "conv1_d" if i == 0 else f"conv1_d_{i}"
        y = conv1d_same(y, P.get(f"{pre}/~/{cv}", "w"), P.get(f"{pre}/~/{cv}", "b"))
        if i < 4:
            bn = "batch_norm" if i == 0 else f"batch_norm_{i}"
            y = batchnorm_eval(
                y,
                P.get(f"{pre}/~/{bn}", "scale"),
                P.get(f"{pre}/~/{bn}", "offset"),
                P.get(f"{pre}/~/{bn}/~/mean_ema", "average", state=True),
                P.get(f"{pre}/~/{bn}/~/var_ema", "average", state=True),
            )
            y = np.tanh(y)
    return out + y


# ---------------------------------------------------------------------------------------------------------------------
# Keep masks as the HIP library draws them (include/vtts_nat.h: vtts_nat_acoustic_keep_masks): Threefry-2x32 with 20 rounds
# (Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11 — Random123's threefry2x32_R(20, ...),
# the block cipher jax.random uses).  Pinned by Random123's known-answer vectors (tests/test_nat_cpu.py).  NOT Haiku's
# key-splitting schedule: only the cipher is shared with the reference.
# ---------------------------------------------------------------------------------------------------------------------
_TF_ROT = (13, 15, 26, 6, 17, 29, 16, 24)


def threefry2x32_20(k0, k1, x0, x1):
    """Vectorised over numpy uint32 arrays (or scalars)."""
    k0, k1, x0, x1 = (np.asarray(v, dtype=np.uint32) for v in (k0, k1, x0, x1))
    ks = (k0, k1, np.uint32(0x1BD11BDA) ^ k0 ^ k1)
    with np.errstate(over="ignore"):
        x0 = x0 + ks[0]
        x1 = x1 + ks[1]
        for g in range(5):
            for r in range(4):
                rot = _TF_ROT[(g & 1) * 4 + r]
                x0 = x0 + x1
                x1 = (x1 << np.uint32(rot)) | (x1 >> np.uint32(32 - rot))
                x1 = x1 ^ x0
            x0 = x0 + ks[(g + 1) % 3]
            x1 = x1 + ks[(g + 2) % 3] + np.uint32(g + 1)
    return x0, x1


def threefry_keep_masks(seed: int, n_frames: int, prenet_dim: int = 256) -> np.ndarray:
    """``[n_frames, 2, prenet_dim]`` boolean keep masks of one sentence: key = seed, counter = (2 * frame + layer,
    64-column block), bit j of (x0 | x1 << 32) = column 64 * block + j."""
    nblk = (prenet_dim + 63) // 64
    ctr0 = np.repeat(np.arange(2 * n_frames, dtype=np.uint32), nblk)
    ctr1 = np.tile(np.arange(nblk, dtype=np.uint32), 2 * n_frames)
    x0, x1 = threefry2x32_20(np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF), ctr0, ctr1)
    bits = np.concatenate([(x0[:, None] >> np.arange(32, dtype=np.uint32)) & 1, (x1[:, None] >> np.arange(32, dtype=np.uint32)) & 1], axis=1)
    return bits.reshape(n_frames, 2, nblk * 64)[:, :, :prenet_dim].astype(bool)


# ---------------------------------------------------------------------------------------------------------------------
# The reference's OWN mask stream (restated round 2): jax.random on the classic (non-partitionable) threefry implementation
# — the default of every JAX release before 0.5 — under dm-haiku's PRNGSequence.  **Unpinned by a JAX run** (no jax
# offline); pinned by the two known answers JAX's own documentation prints for PRNGKey(0) (tests/test_nat_cpu.py).
#
#   jax.random.split(key, 2)          counts = iota(uint32, 4); threefry_2x32 splits the counts into halves x0 = [0, 1],
#       (jax/_src/prng.py)            x1 = [2, 3], enciphers the pairs (x0[i], x1[i]) with the key and concatenates the two
#                                     output halves: new keys = [[y0[0], y0[1]], [y1[0], y1[1]]]
#   jax.random.bits / uniform         counts = iota(uint32, n) (n = number of 32-bit words; an odd n is padded with one 0),
#                                     halves as above; word i < n/2 = y0[i], word n/2 + i = y1[i]; uniform = bitcast(
#                                     (word >> 9) | 0x3F800000) - 1.0
#   jax.random.bernoulli(key, p, s)   uniform(key, s) < p
#   hk.dropout(rng, rate, x)          keep = bernoulli(rng, 1 - rate, x.shape); keep * x / (1 - rate)      (haiku/_src/basic.py)
#   hk.next_rng_key()                 PRNGSequence: (key, sub) = split(key, 2); key stays, sub is handed out — one split per
#                                     call (reserve size 1).  hk.scan threads the sequence's state through the scan carry;
#                                     releases that reserve a subkey before the scan and after every step consume the SAME
#                                     chain of subkeys, only earlier.
#   AcousticModel.inference           the only consumers are the prenet's two dropouts per frame (model.py:95-100,134-142;
#                                     encoder and postnet draw nothing with is_training=False): with K_0 = the
#                                     checkpoint's rng (text2mel.py:65-73) and (K_n, S_n) = split(K_{n-1}), frame t takes
#                                     S_{2t+1} for the first mask and S_{2t+2} for the second, shape (1, 256): every
#                                     sentence starts from the same K_0.
# JAX >= 0.5 defaults to jax_threefry_partitionable=True (another counter layout for split and bits): a checkpoint run
# there draws a different stream from the same key.  The reference pins no version (setup.py:6-19).
# ---------------------------------------------------------------------------------------------------------------------
def jax_legacy_threefry_2x32(key, counts):
    """jax._src.prng.threefry_2x32 (classic layout): ``counts`` uint32 [n] -> uint32 [n]."""
    counts = np.asarray(counts, dtype=np.uint32).ravel()
    odd = counts.size % 2
    if odd:
        counts = np.concatenate([counts, np.zeros(1, np.uint32)])
    half = counts.size // 2
    y0, y1 = threefry2x32_20(np.uint32(key[0]), np.uint32(key[1]), counts[:half], counts[half:])
    out = np.concatenate([y0, y1])
    return out[:-1] if odd else out


def jax_legacy_split(key, num: int = 2) -> np.ndarray:
    """jax.random.split(key, num) -> uint32 [num, 2]."""
    return jax_legacy_threefry_2x32(key, np.arange(2 * num, dtype=np.uint32)).reshape(num, 2)


def jax_legacy_uniform(key, n: int) -> np.ndarray:
    """jax.random.uniform(key, (n,), float32) in [0, 1)."""
    bits = jax_legacy_threefry_2x32(key, np.arange(n, dtype=np.uint32))
    return ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)


# The same two primitives under ``jax_threefry_partitionable=True`` (the default from JAX 0.5 on; jax/_src/prng.py:
# ``_threefry_split_foldlike`` and ``_threefry_random_bits_partitionable``), restated FROM RECOLLECTION of that file — no known
# answer for this mode is quotable from memory and no JAX runs here, so this mode is **unpinned** (tests check device == this
# restatement and that the two modes differ; nothing checks this restatement against JAX):
#   split(key, n)[i]   = threefry2x32(key, (hi, lo) of the 64-bit index i)            -> the pair (y0, y1) IS subkey i
#   bits(key, shape)   : element i (row-major index, 64 bit) -> threefry2x32(key, (hi(i), lo(i))), word = y0 ^ y1 (32-bit draws)
def jax_partitionable_split(key, num: int = 2) -> np.ndarray:
    idx = np.arange(num, dtype=np.uint32)
    y0, y1 = threefry2x32_20(np.uint32(key[0]), np.uint32(key[1]), np.zeros(num, np.uint32), idx)
    return np.stack([y0, y1], axis=1)


def jax_partitionable_uniform(key, n: int) -> np.ndarray:
    idx = np.arange(n, dtype=np.uint32)
    y0, y1 = threefry2x32_20(np.uint32(key[0]), np.uint32(key[1]), np.zeros(n, np.uint32), idx)
    bits = y0 ^ y1
    return ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)


def haiku_prenet_keep_masks(rng_key, n_frames: int, prenet_dim: int = 256, partitionable: bool = False) -> np.ndarray:
    """``[n_frames, 2, prenet_dim]`` boolean keep masks of AcousticModel.inference's prenet dropout as the reference draws
    them from ``rng_key`` (uint32 [2]: the checkpoint's ``rng``) — see the block comment above.  ``partitionable``: the layout of
    ``jax_threefry_partitionable=True`` (JAX >= 0.5's default; unpinned, see above) instead of the classic one."""
    key = np.asarray(rng_key, dtype=np.uint32).reshape(2)
    split, uniform = (jax_partitionable_split, jax_partitionable_uniform) if partitionable else (jax_legacy_split, jax_legacy_uniform)
    out = np.empty((n_frames, 2, prenet_dim), dtype=bool)
    for t in range(n_frames):
        for layer in range(2):
            key, sub = split(key, 2)
            out[t, layer] = uniform(sub, prenet_dim) < np.float32(0.5)
    return out
