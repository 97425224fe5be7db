"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A numpy stand-in for the slice of dm-haiku / jax that the reference's NAT code touches, so that the REFERENCE'S OWN FILES
(vietTTS/nat/model.py, vietTTS/nat/text2mel.py, imported from where they lie under /root/reference) execute here, where
jax / jaxlib / dm-haiku cannot be installed.  What this buys: the model WIRING — which layers in which order, the masks,
concatenations, the order of ``hk.next_rng_key()`` draws, the silence rules and frame arithmetic of ``text2mel`` — is then
the reference's source, executed, not a restatement; ``oracle/make_nat_golden.py`` mints golden vectors from it and
``oracle/nat_oracle.py`` / the HIP path are checked against those.

What it does NOT buy: the primitives below (hk.LSTM, hk.Conv1D, hk.BatchNorm, hk.ResetCore, hk.dynamic_unroll,
hk.deep_rnn_with_skip_connections, hk.dropout + the PRNG key chain, module naming) are this repo's reading of the
third-party libraries (the same functions as oracle/nat_oracle.py, whose header lists the sources) — **unpinned by a JAX
run**.  A wrong reading of a primitive is shared by shim and oracle and is not caught here; since round 5 the ARITHMETIC of each
(LSTM step and sequence, BatchNorm in eval mode, tanh-gelu, softplus, SAME convolution, the upsampling softmax) is pinned against
PyTorch's independent implementations (tests/test_nat_primitives_torch_cpu.py, 1e-12 in float64); Haiku's conventions (gate order,
forget bias, layouts, module naming, the rng key chain) stay by reading.

Module naming follows Haiku's rules: snake_case class name, numbered per creating scope (``lstm``, ``lstm_1``), joined
with ``/~/`` when created inside the parent's ``__init__`` and ``/`` when created inside another method.

Usage (oracle/make_nat_golden.py): ``install()`` puts ``haiku``, ``jax``, ``jax.numpy``, ``jax.nn``, ``jax.random`` (and an empty
``textgrid``) into ``sys.modules``; ``hk.transform_with_state(f).apply(params, state, rng, *args)`` runs ``f`` against plain
dicts of numpy arrays; ``set_dtype(np.float64)`` makes every parameter / state array arrive in that precision.
"""
from __future__ import annotations

import re
import sys
import types
from collections import namedtuple
from typing import Dict, List, Optional

import numpy as np

from . import hifigan_oracle as H
from . import nat_oracle as O

_DTYPE = [np.float64]


def set_dtype(dt) -> None:
    _DTYPE[0] = dt


# ---------------------------------------------------------------------------------------------------------------------
# frame: what hk.transform_with_state(...).apply installs
# ---------------------------------------------------------------------------------------------------------------------
class _Frame:
    def __init__(self, params, state, rng):
        self.params, self.state = params, state
        self.key = None if rng is None else np.asarray(rng, dtype=np.uint32).reshape(2)
        self.stack: List[tuple] = []          # (module, method name, name counters of THIS method invocation)
        self.top_counters: Dict[str, int] = {}
        self.rng_draws = 0


_FRAMES: List[_Frame] = []


def _frame() -> _Frame:
    if not _FRAMES:
        raise RuntimeError("haiku shim: used outside transform_with_state(...).apply")
    return _FRAMES[-1]


def _snake(name: str) -> str:
    """Haiku's utils.camel_to_snake: Conv1D -> conv1_d, LSTM -> lstm, BatchNorm -> batch_norm."""
    s = re.sub(r"((?<=[a-z0-9])[A-Z]|(?!^)[A-Z](?=[a-z]))", r"_\1", name)
    return s.lower()


class _ModuleMeta(type):
    """Names the instance from the CREATING scope before its __init__ runs, and runs every method under the module's scope."""

    def __new__(mcs, name, bases, ns):
        for k, v in list(ns.items()):
            if callable(v) and not isinstance(v, (staticmethod, classmethod, type)) and (k in ("__call__",) or not k.startswith("_")):
                ns[k] = _scoped(k, v)
        return super().__new__(mcs, name, bases, ns)

    def __call__(cls, *args, **kwargs):
        fr = _frame()
        obj = cls.__new__(cls)
        fr.stack.append((obj, "__init__", {}))
        try:
            obj.__init__(*args, **kwargs)  # Module.__init__ (reached through super().__init__(name=...)) names the instance
            if not hasattr(obj, "module_name"):
                raise RuntimeError(f"{cls.__name__}.__init__ never called super().__init__() (Haiku raises here too)")
        finally:
            fr.stack.pop()
        return obj


def _scoped(method_name, fn):
    def wrapper(self, *a, **k):
        fr = _frame()
        fr.stack.append((self, method_name, {}))
        try:
            return fn(self, *a, **k)
        finally:
            fr.stack.pop()

    wrapper.__name__ = getattr(fn, "__name__", method_name)
    return wrapper


class Module(metaclass=_ModuleMeta):
    def __init__(self, name: Optional[str] = None):
        """Haiku names the module HERE, from the scope that is creating it: ``<parent>/~/<name>`` inside the parent's ``__init__``,
        ``<parent>/<name>`` inside another of its methods; siblings of one method INVOCATION are numbered (a fresh counter per
        wrapped call: the hk.Linear that hk.LSTM.__call__ builds is "linear" at every step; the three hk.Conv1D of one __init__ are
        conv1_d, conv1_d_1, conv1_d_2)."""
        fr = _frame()
        assert fr.stack and fr.stack[-1][0] is self, "Module.__init__ outside its constructor"
        base = name or _snake(type(self).__name__)
        if len(fr.stack) >= 2:
            parent, method, cnt = fr.stack[-2]
            prefix = parent.module_name + ("/~/" if method == "__init__" else "/")
        else:
            prefix, cnt = "", fr.top_counters
        n = cnt.get(base, 0)
        cnt[base] = n + 1
        self.module_name = prefix + (base if n == 0 else f"{base}_{n}")


def _current_module() -> Module:
    fr = _frame()
    if not fr.stack:
        raise RuntimeError("haiku shim: parameter requested outside a module")
    return fr.stack[-1][0]


def get_parameter(name: str) -> np.ndarray:
    fr, m = _frame(), _current_module()
    try:
        return np.asarray(fr.params[m.module_name][name]).astype(_DTYPE[0])
    except KeyError:
        raise KeyError(f"haiku shim: no parameter {m.module_name!r} / {name!r} in the checkpoint (have {sorted(fr.params)[:4]} ...)") from None


def _get_state(module_name: str, name: str) -> np.ndarray:
    return np.asarray(_frame().state[module_name][name]).astype(_DTYPE[0])


def set_state(name: str, value) -> None:  # model.py:109 stores the attention map for plotting: not an output
    return None


def next_rng_key() -> np.ndarray:
    """hk.PRNGSequence: reserve(1) = ``new_key, subkey = jax.random.split(key, 2)``; the subkey is handed out."""
    fr = _frame()
    if fr.key is None:
        raise RuntimeError("haiku shim: next_rng_key() without an rng")
    ks = O.jax_legacy_split(fr.key, 2)
    fr.key = ks[0]
    fr.rng_draws += 1
    return ks[1]


def dropout(rng, rate: float, x):
    """hk.dropout: keep = jax.random.bernoulli(rng, 1 - rate, x.shape) = uniform(rng, x.shape) < 1 - rate; x * keep / (1 - rate)."""
    x = np.asarray(x)
    keep_rate = 1.0 - rate
    keep = (O.jax_legacy_uniform(rng, x.size) < np.float32(keep_rate)).reshape(x.shape)
    return np.where(keep, x / x.dtype.type(keep_rate), x.dtype.type(0))


# ---------------------------------------------------------------------------------------------------------------------
# layers (primitive arithmetic = oracle/nat_oracle.py)
# ---------------------------------------------------------------------------------------------------------------------
class Embed(Module):
    def __init__(self, vocab_size, embed_dim, name=None):
        super().__init__(name=name)

    def __call__(self, ids):
        return get_parameter("embeddings")[np.asarray(ids)]


class Linear(Module):
    def __init__(self, output_size, with_bias=True, name=None):
        super().__init__(name=name)
        self.with_bias = with_bias

    def __call__(self, x):
        y = np.asarray(x) @ get_parameter("w")
        return y + get_parameter("b") if self.with_bias else y


class Conv1D(Module):
    """hk.Conv1D(output_channels, kernel_shape, stride=1, rate=1, padding="SAME" | ((lo, hi),)): NWC, w [k, Cin, Cout], cross-correlation."""

    def __init__(self, output_channels, kernel_shape, stride=1, rate=1, padding="SAME", name=None):
        super().__init__(name=name)
        assert stride == 1
        self.rate, self.padding = rate, padding

    def __call__(self, x):  # [B, L, C]
        w, b = get_parameter("w"), get_parameter("b")
        if self.padding == "SAME":
            assert self.rate == 1
            return np.stack([O.conv1d_same(np.asarray(xb), w, b) for xb in x])
        (lo, hi), = self.padding
        assert lo == hi
        return H.conv1d(np.asarray(x), w, b, self.rate, lo)


class Conv1DTranspose(Module):
    """hk.Conv1DTranspose(output_channels, kernel_shape, stride, padding="SAME"): w [k, Cout, Cin] (oracle/hifigan_oracle.py)."""

    def __init__(self, output_channels, kernel_shape=None, stride=1, padding="SAME", name=None):
        super().__init__(name=name)
        assert padding == "SAME"
        self.stride = stride

    def __call__(self, x):
        return H.conv1d_transpose(np.asarray(x), get_parameter("w"), get_parameter("b"), self.stride)


class _Ema(Module):
    pass


class BatchNorm(Module):
    def __init__(self, create_scale, create_offset, decay_rate, name=None):
        super().__init__(name=name)
        self.mean_ema = _Ema(name="mean_ema")
        self.var_ema = _Ema(name="var_ema")

    def __call__(self, x, is_training):
        assert not is_training, "the shim runs inference only"
        return O.batchnorm_eval(np.asarray(x), get_parameter("scale"), get_parameter("offset"),
                                _get_state(self.mean_ema.module_name, "average"), _get_state(self.var_ema.module_name, "average"))


LSTMState = namedtuple("LSTMState", ["hidden", "cell"])


class RNNCore(Module):
    pass


class LSTM(RNNCore):
    def __init__(self, hidden_size, name=None):
        super().__init__(name=name)
        self.hidden_size = hidden_size

    def initial_state(self, batch_size):
        z = np.zeros((batch_size, self.hidden_size), _DTYPE[0])
        return LSTMState(hidden=z, cell=z.copy())

    def __call__(self, inputs, prev_state):
        lin = _LstmLinear(name="linear")  # hk.LSTM builds hk.Linear(4 * hidden) inside __call__: "<lstm>/linear"
        w, b = lin.wb()
        h, c = O.lstm_step(np.asarray(inputs), prev_state.hidden, prev_state.cell, w, b)
        return h, LSTMState(hidden=h, cell=c)


class _LstmLinear(Module):
    def wb(self):
        return get_parameter("w"), get_parameter("b")


class ResetCore(RNNCore):
    """hk.ResetCore: ``state = where(should_reset, initial_state, state)`` BEFORE the wrapped core's step."""

    def __init__(self, core, name=None):
        super().__init__(name=name)
        self.core = core

    def initial_state(self, batch_size):
        return self.core.initial_state(batch_size)

    def __call__(self, inputs, state):
        x, should_reset = inputs
        init = self.core.initial_state(np.asarray(x).shape[0])
        m = np.asarray(should_reset).astype(bool)[:, None]
        state = tree_map(lambda i, s: np.where(m, i, s), init, state)
        return self.core(x, state)


class _DeepRNN(RNNCore):
    """dm-haiku recurrent.py::_DeepRNN with skip_connections=True: layer idx > 0 sees concat([inputs, previous output]);
    the output is the concatenation of every layer's output."""

    def __init__(self, layers, name=None):
        super().__init__(name=name)
        self.layers = list(layers)

    def initial_state(self, batch_size):
        return tuple(l.initial_state(batch_size) for l in self.layers)

    def __call__(self, inputs, state):
        cur, outs, nxt = inputs, [], []
        for idx, layer in enumerate(self.layers):
            if idx > 0:
                cur = np.concatenate([inputs, cur], axis=-1)
            cur, s = layer(cur, state[idx])
            outs.append(cur)
            nxt.append(s)
        return np.concatenate(outs, axis=-1), tuple(nxt)


def deep_rnn_with_skip_connections(layers, name=None):
    return _DeepRNN(layers, name=name or "deep_rnn")


class Sequential(Module):
    def __init__(self, layers, name=None):
        super().__init__(name=name)
        self.layers = list(layers)

    def __call__(self, x):
        for l in self.layers:
            x = l(x)
        return x


def dynamic_unroll(core, input_sequence, initial_state, time_major=True):
    """hk.dynamic_unroll = hk.scan over time with Haiku's internal state (the rng key included) threaded through the carry:
    the same sequence of effects as this loop."""
    axis = 0 if time_major else 1
    leaves = _leaves(input_sequence)
    T = np.asarray(leaves[0]).shape[axis]
    state, outs = initial_state, []
    for t in range(T):
        xt = tree_map(lambda a: np.take(np.asarray(a), t, axis=axis), input_sequence)
        y, state = core(xt, state)
        outs.append(y)
    return tree_map(lambda *ys: np.stack(ys, axis=axis), *outs), state


class _Transformed:
    def __init__(self, fn):
        self.fn = fn
        self.last_rng_draws = 0

    def apply(self, params, state, rng, *args, **kwargs):
        fr = _Frame(params, state, rng)
        _FRAMES.append(fr)
        try:
            out = self.fn(*args, **kwargs)
        finally:
            _FRAMES.pop()
        self.last_rng_draws = fr.rng_draws
        return out, state


def transform_with_state(fn):
    return _Transformed(fn)


class PRNGSequence:
    """hk.PRNGSequence(seed or key): ``next`` = reserve(1) = split(key, 2), hand out the second."""

    def __init__(self, key_or_seed):
        self.key = (np.array([0, int(key_or_seed) & 0xFFFFFFFF], np.uint32) if np.isscalar(key_or_seed)
                    else np.asarray(key_or_seed, np.uint32).reshape(2))

    def __iter__(self):
        return self

    def __next__(self):
        ks = O.jax_legacy_split(self.key, 2)
        self.key = ks[0]
        return ks[1]


# ---------------------------------------------------------------------------------------------------------------------
# jax
# ---------------------------------------------------------------------------------------------------------------------
def _is_leaf(x) -> bool:
    return x is None or not isinstance(x, (tuple, list, dict))


def _leaves(tree) -> list:
    if _is_leaf(tree):
        return [tree]
    if isinstance(tree, dict):
        return [l for k in sorted(tree) for l in _leaves(tree[k])]
    return [l for t in tree for l in _leaves(t)]


def tree_map(f, tree, *rest):
    if tree is None:
        return None
    if _is_leaf(tree):
        return f(tree, *rest)
    if isinstance(tree, dict):
        return {k: tree_map(f, tree[k], *[r[k] for r in rest]) for k in tree}
    mapped = [tree_map(f, t, *[r[i] for r in rest]) for i, t in enumerate(tree)]
    if hasattr(tree, "_fields"):
        return type(tree)(*mapped)
    return type(tree)(mapped)


def _softmax(x, axis=-1):
    x = np.asarray(x)
    z = x - x.max(axis=axis, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=axis, keepdims=True)


def _jit(f, static_argnums=None, **_):
    return f


def _device_get(x):
    return x


def _build_modules() -> Dict[str, types.ModuleType]:
    hk = types.ModuleType("haiku")
    for k, v in dict(Module=Module, Embed=Embed, Linear=Linear, Conv1D=Conv1D, BatchNorm=BatchNorm, LSTM=LSTM, LSTMState=LSTMState, RNNCore=RNNCore,
                     ResetCore=ResetCore, Sequential=Sequential, Conv1DTranspose=Conv1DTranspose, PRNGSequence=PRNGSequence, dropout=dropout, next_rng_key=next_rng_key, set_state=set_state,
                     dynamic_unroll=dynamic_unroll, deep_rnn_with_skip_connections=deep_rnn_with_skip_connections,
                     transform_with_state=transform_with_state).items():
        setattr(hk, k, v)

    jnp = types.ModuleType("jax.numpy")
    for k in ("arange", "flip", "concatenate", "squeeze", "cumsum", "square", "einsum", "zeros", "array", "tanh", "where", "sum", "int32", "float32", "ndarray",
              "exp", "log", "maximum", "minimum", "stack", "ones", "mean", "abs", "sqrt", "asarray", "float64", "int64"):
        setattr(jnp, k, getattr(np, k))
    jnp.clip = lambda a, a_min=None, a_max=None: np.clip(a, a_min, a_max)

    nn = types.ModuleType("jax.nn")
    nn.relu = lambda x: np.maximum(np.asarray(x), 0)
    nn.gelu = lambda x, approximate=True: O.gelu_tanh(np.asarray(x))
    nn.softplus = lambda x: O.softplus(np.asarray(x))
    nn.softmax = _softmax
    nn.leaky_relu = lambda x, negative_slope=0.01: H.leaky_relu(np.asarray(x), negative_slope)

    rnd = types.ModuleType("jax.random")
    rnd.split = lambda key, num=2: O.jax_legacy_split(np.asarray(key, np.uint32), num)
    rnd.PRNGKey = lambda seed: np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], np.uint32)

    jax = types.ModuleType("jax")
    jax.numpy, jax.nn, jax.random = jnp, nn, rnd
    jax.tree_map = tree_map
    jax.jit = _jit
    jax.device_get = _device_get
    return {"haiku": hk, "jax": jax, "jax.numpy": jnp, "jax.nn": nn, "jax.random": rnd, "textgrid": types.ModuleType("textgrid")}


def install() -> Dict[str, types.ModuleType]:
    """Put the stand-ins into sys.modules (refuses to shadow a real jax / haiku)."""
    for name in ("jax", "haiku"):
        m = sys.modules.get(name)
        if m is not None and not getattr(m, "__vtts_shim__", False):
            raise RuntimeError(f"a real {name} is already imported: run the reference on it instead of this shim")
    mods = _build_modules()
    for name, m in mods.items():
        m.__vtts_shim__ = True
        sys.modules[name] = m
    return mods


def uninstall() -> None:
    for name in ("haiku", "jax", "jax.numpy", "jax.nn", "jax.random", "textgrid"):
        m = sys.modules.get(name)
        if m is not None and getattr(m, "__vtts_shim__", False):
            del sys.modules[name]
