"""Reading the reference's NAT checkpoints where jax / dm-haiku / optax are NOT installed.

The reference pickles its training state as it is (vietTTS/nat/utils.py:18-26: ``{"step", "params", "aux", "rng", "optim_state"}``
with Haiku mappings of jax arrays and optax state tuples inside), so ``pickle.load`` of a real ``*_latest_ckpt.pickle`` imports those
libraries (vietTTS/nat/text2mel.py:27-28, :62-71).  On an MI355X serving box they are not there.  This loader reads such a file
anyway, best effort:

* a class or function of an absent library (``haiku``, ``jax``, ``jaxlib``, ``optax``, ``chex``, ``flax``) unpickles as an inert
  stand-in that only records its constructor arguments and state;
* jax arrays: both array types reduce to ``(reconstruct, (numpy_reduce_fn, numpy_reduce_args, numpy_state, aval_state))``
  (``jax._src.device_array.reconstruct_device_array`` before 0.4, ``jax._src.array._reconstruct_array`` since) — rebuilt here as the
  plain ``numpy.ndarray`` they wrap;
* mappings (Haiku's ``FlatMapping``, flax ``FrozenDict`` ...): the stand-in is unwrapped to the one mapping found among its recorded
  arguments / state; a layout this cannot recognise raises :class:`CheckpointFormatError` naming the class, instead of guessing.

**Unverified against a real checkpoint** (none ships with the reference; its download is not reachable offline): the formats above are
restated from the libraries' pickling code and exercised by tests on emulated classes (tests/test_nat_cpu.py).  The certain route is
to re-save once where jax + haiku exist: plain dicts of numpy arrays load with no special handling.
"""
from __future__ import annotations

import pickle
from collections.abc import Mapping
from typing import Any, Dict

import numpy as np

ABSENT_ROOTS = ("haiku", "jax", "jaxlib", "optax", "chex", "flax", "ml_dtypes")


class CheckpointFormatError(ValueError):
    pass


class _Absent:
    """Stand-in for an instance of a class whose library is not installed: records, never computes."""

    _vtts_origin = "?"
    _args, _kwargs, _state = (), {}, None  # defaults for instances pickle creates through copyreg._reconstructor (protocol < 2)

    def __new__(cls, *args, **kwargs):
        self = object.__new__(cls)
        self._args, self._kwargs, self._state = args, kwargs, None
        return self

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        self._state = state

    def __repr__(self):
        return f"<absent {self._vtts_origin}>"


def _absent_class(module: str, name: str):
    return type(name, (_Absent,), {"__module__": module, "_vtts_origin": f"{module}.{name}"})


def _rebuild_numpy(fun, args, arr_state, *_aval_state):
    """jax's array ``__reduce__``: ``fun(*args)`` is numpy's empty reconstructor, ``arr_state`` the ndarray's pickled state."""
    value = fun(*args)
    value.__setstate__(arr_state)
    return np.asarray(value)


class TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            root = module.split(".")[0]
            if root not in ABSENT_ROOTS:
                raise
            if name in ("reconstruct_device_array", "_reconstruct_array"):
                return _rebuild_numpy
            return _absent_class(module, name)


def to_plain(obj: Any) -> Any:
    """Strip the stand-ins: mappings -> dict, arrays -> numpy, sequences keep their kind."""
    if isinstance(obj, np.ndarray) or obj is None or isinstance(obj, (bool, int, float, str, bytes, np.generic)):
        return obj
    if isinstance(obj, Mapping):
        return {k: to_plain(v) for k, v in obj.items()}
    if isinstance(obj, _Absent):
        pool = list(obj._args) + list(obj._kwargs.values())
        if isinstance(obj._state, Mapping):
            pool += list(obj._state.values())
        elif obj._state is not None:
            pool.append(obj._state)
        maps = [p for p in pool if isinstance(p, Mapping)]
        if len(maps) == 1:
            return to_plain(maps[0])
        arrs = [p for p in pool if isinstance(p, np.ndarray)]
        if len(arrs) == 1 and not maps:
            return arrs[0]
        raise CheckpointFormatError(
            f"cannot recover the content of a pickled {obj._vtts_origin} without its library ({len(maps)} mappings, {len(arrs)} arrays among its "
            "recorded arguments): re-save the checkpoint as plain numpy dicts where jax + haiku are installed (INTEGRATION.md section 6)")
    if isinstance(obj, tuple) and hasattr(obj, "_fields"):
        return type(obj)(*[to_plain(v) for v in obj])
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_plain(v) for v in obj)
    return obj


def load_checkpoint(path) -> Dict[str, Any]:
    """``{"step", "params", "aux", "rng", ...}`` with ``params`` / ``aux`` as ``{module: {name: ndarray}}`` and ``rng`` as an ndarray (or
    None); ``optim_state`` is left as unpickled (inference never reads it)."""
    with open(path, "rb") as f:
        dic = TolerantUnpickler(f).load()
    if not isinstance(dic, Mapping) or "params" not in dic:
        raise CheckpointFormatError(f"{path}: not a vietTTS checkpoint (expected a dict with 'params', 'aux', 'rng')")
    out = dict(dic)
    for k in ("params", "aux"):
        plain = to_plain(dic.get(k) or {})
        out[k] = {str(m): {str(n): np.asarray(a) for n, a in dict(v).items()} for m, v in dict(plain).items()}
    rng = to_plain(dic.get("rng"))
    out["rng"] = None if rng is None else np.asarray(rng)
    _validate(path, out)
    return out


def _validate(path, out) -> None:
    """A recovered (possibly guessed, see to_plain) layout is accepted only if it LOOKS like a Haiku parameter tree: module names as Haiku
    forms them, every leaf a real-valued array, the rng a 2-word key.  Names and shapes are then checked one by one against the
    model's own parameter table when they are handed to the C ABI (set_param rejects unknown modules and wrong shapes, pack() a
    missing array): nothing unrecognised gets as far as a forward pass."""
    if not out["params"]:
        raise CheckpointFormatError(f"{path}: 'params' is empty after unwrapping")
    for k in ("params", "aux"):
        for mod, leaves in out[k].items():
            if not leaves or "/" not in mod and "~" not in mod and not mod.replace("_", "").isalnum():
                raise CheckpointFormatError(f"{path}: {k}[{mod!r}] does not look like a Haiku module entry")
            for name, arr in leaves.items():
                if not isinstance(arr, np.ndarray) or arr.dtype.kind not in "fiu" or arr.dtype == object:
                    raise CheckpointFormatError(f"{path}: {k}[{mod!r}][{name!r}] is not a numeric array ({type(arr).__name__}, dtype {getattr(arr, 'dtype', None)})")
                if arr.dtype.kind == "f" and not np.isfinite(arr).all():
                    raise CheckpointFormatError(f"{path}: {k}[{mod!r}][{name!r}] holds non-finite values")
    rng = out["rng"]
    if rng is not None and not (rng.shape == (2,) and rng.dtype.kind in "ui"):
        raise CheckpointFormatError(f"{path}: 'rng' is not a jax.random.PRNGKey (uint32[2]): shape {rng.shape}, dtype {rng.dtype}")
