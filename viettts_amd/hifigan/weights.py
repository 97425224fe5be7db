"""Weight formats on either side of the hot path.

Two on-disk formats exist in the reference:

* ``hk_hifi.pickle`` — ``dict[module_name] -> {"w": ndarray, "b": ndarray}`` read by
  ``mel2wave`` (vietTTS/hifigan/mel2wave.py:35-36).  Layouts:
  conv ``w[K, Cin, Cout]``, transposed conv ``w[K, Cout, Cin]`` with the K axis already
  flipped relative to PyTorch, biases ``b[Cout]``.
* upstream ``jik876/hifi-gan`` generator state dicts (``g_XXXXXXXX``), converted by
  vietTTS/hifigan/convert_torch_model_to_haiku.py:27-62.

This module restates the converter's *name map* (:36-46) and *layout map* (:50-58) as pure
functions so that both formats can feed the engine, and enumerates the 78 convolution
modules of the generator in execution order (vietTTS/hifigan/model.py:78-107).
"""
from __future__ import annotations

import pickle
from dataclasses import dataclass
from typing import Dict, Iterator, List, Mapping

import numpy as np

from .config import HifiganConfig

ParamDict = Dict[str, Dict[str, np.ndarray]]


@dataclass(frozen=True)
class ConvSpec:
    """One convolution module of the generator."""

    key: str  # Haiku module name (pickle key)
    torch_prefix: str  # upstream state-dict prefix, e.g. "resblocks.4.convs1.2"
    kind: str  # "conv" | "convT"
    cin: int
    cout: int
    k: int
    dilation: int = 1
    stride: int = 1

    @property
    def w_shape(self):
        """Shape in the Haiku pickle."""
        if self.kind == "conv":
            return (self.k, self.cin, self.cout)
        return (self.k, self.cout, self.cin)

    @property
    def torch_w_shape(self):
        if self.kind == "conv":
            return (self.cout, self.cin, self.k)
        return (self.cin, self.cout, self.k)


def conv_specs(cfg: HifiganConfig) -> List[ConvSpec]:
    """All convolution modules in the order ``Generator.__call__`` executes them
    (vietTTS/hifigan/model.py:109-125), which is also upstream's ``named_modules()`` order
    except that conv_pre comes first and conv_post last in both."""
    cfg.validate()
    specs: List[ConvSpec] = []
    c0 = cfg.upsample_initial_channel
    # model.py:83 -> first hk.Conv1D gets the default name "conv1_d"
    specs.append(ConvSpec("generator/~/conv1_d", "conv_pre", "conv", cfg.num_mels, c0, 7))
    ups, res = [], []
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        cin = c0 // (2 ** i)
        cout = c0 // (2 ** (i + 1))
        ups.append(ConvSpec(f"generator/~/ups_{i}", f"ups.{i}", "convT", cin, cout, int(k), 1, int(u)))
    n = 0
    for i in range(cfg.num_upsamples):
        ch = cfg.stage_channels(i)
        for k, dil in zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes):
            blk = []
            if cfg.resblock == "2":
                # ResBlock2 (model.py:54-74): the module keeps the name res_block1_N (model.py:105) and its two hk.Conv1D get the
                # default names conv1_d, conv1_d_1 (model.py:58-66); upstream torch: resblocks.N.convs.Z (torch_model.py:119-141)
                for z in range(2):
                    blk.append(ConvSpec(f"generator/~/res_block1_{n}/~/conv1_d" + ("" if z == 0 else f"_{z}"), f"resblocks.{n}.convs.{z}", "conv",
                                        ch, ch, int(k), int(dil[z])))
                res.append(blk)
                n += 1
                continue
            for z in range(3):  # ResBlock1.__call__ runs convs1_z then convs2_z (model.py:45-49)
                blk.append(
                    ConvSpec(f"generator/~/res_block1_{n}/~/convs1_{z}", f"resblocks.{n}.convs1.{z}", "conv", ch, ch, int(k), int(dil[z]))
                )
                blk.append(
                    ConvSpec(f"generator/~/res_block1_{n}/~/convs2_{z}", f"resblocks.{n}.convs2.{z}", "conv", ch, ch, int(k), 1)
                )
            res.append(blk)
            n += 1
    # execution order: ups_i followed by its three resblocks
    nk = cfg.num_kernels
    for i in range(cfg.num_upsamples):
        specs.append(ups[i])
        for j in range(nk):
            specs.extend(res[i * nk + j])
    # model.py:107 -> second default-named hk.Conv1D is "conv1_d_1"
    specs.append(ConvSpec("generator/~/conv1_d_1", "conv_post", "conv", cfg.stage_channels(cfg.num_upsamples - 1), 1, 7))
    return specs


def specs_named_modules_order(cfg: HifiganConfig) -> List[ConvSpec]:
    """Upstream ``Generator.named_modules()`` order: conv_pre, ups.*, resblocks.*, conv_post
    (vietTTS/hifigan/torch_model.py:162-189).  Used only to draw synthetic weights in a
    documented order."""
    s = conv_specs(cfg)
    pre = [x for x in s if x.torch_prefix == "conv_pre"]
    ups = [x for x in s if x.kind == "convT"]
    res = [x for x in s if x.torch_prefix.startswith("resblocks.")]
    res.sort(key=lambda x: (int(x.torch_prefix.split(".")[1]), x.torch_prefix.split(".")[2], int(x.torch_prefix.split(".")[3])))
    post = [x for x in s if x.torch_prefix == "conv_post"]
    return pre + ups + res + post


# ---------------------------------------------------------------------------
# layout map (convert_torch_model_to_haiku.py:50-58)
# ---------------------------------------------------------------------------
def torch_weight_to_haiku(spec: ConvSpec, w: np.ndarray) -> np.ndarray:
    """Conv: ``[Cout,Cin,K] -> [K,Cin,Cout]`` (swapaxes(0,2), :55-56).
    Transposed conv: ``[Cin,Cout,K] -> rot90(w, 1, axes=(0,2))`` = ``[K,Cout,Cin]`` with
    ``w_hk[j,o,i] = w[i,o,K-1-j]`` (:53-54)."""
    w = np.asarray(w)
    if tuple(w.shape) != tuple(spec.torch_w_shape):
        raise ValueError(f"{spec.torch_prefix}: expected torch weight {spec.torch_w_shape}, got {w.shape}")
    if spec.kind == "conv":
        return np.ascontiguousarray(np.transpose(w, (2, 1, 0)))
    return np.ascontiguousarray(np.transpose(w[:, :, ::-1], (2, 1, 0)))


def haiku_weight_to_torch(spec: ConvSpec, w: np.ndarray) -> np.ndarray:
    """Inverse of :func:`torch_weight_to_haiku`."""
    w = np.asarray(w)
    if tuple(w.shape) != tuple(spec.w_shape):
        raise ValueError(f"{spec.key}: expected haiku weight {spec.w_shape}, got {w.shape}")
    if spec.kind == "conv":
        return np.ascontiguousarray(np.transpose(w, (2, 1, 0)))
    return np.ascontiguousarray(np.transpose(w, (2, 1, 0))[:, :, ::-1])


def state_dict_to_haiku(cfg: HifiganConfig, state: Mapping[str, np.ndarray]) -> ParamDict:
    """Upstream generator state dict (weight-norm already folded: ``<prefix>.weight`` /
    ``<prefix>.bias``) -> Haiku pickle dict (convert_torch_model_to_haiku.py:33-58).
    Weight-norm checkpoints (``weight_g``/``weight_v``) are folded here the way
    ``remove_weight_norm`` does: ``w = v * (g / ||v||)`` with the norm over all dims but 0."""
    out: ParamDict = {}
    for spec in conv_specs(cfg):
        p = spec.torch_prefix
        if p + ".weight" in state:
            w = np.asarray(_to_numpy(state[p + ".weight"]), dtype=np.float32)
        elif p + ".weight_v" in state:
            v = np.asarray(_to_numpy(state[p + ".weight_v"]), dtype=np.float32)
            g = np.asarray(_to_numpy(state[p + ".weight_g"]), dtype=np.float32)
            norm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(1, 2), keepdims=True)).astype(np.float32)
            w = v * (g / norm)
        else:
            raise KeyError(f"state dict has neither {p}.weight nor {p}.weight_v")
        b = np.asarray(_to_numpy(state[p + ".bias"]), dtype=np.float32)
        out[spec.key] = {"w": torch_weight_to_haiku(spec, w), "b": np.ascontiguousarray(b)}
    return out


def haiku_to_state_dict(cfg: HifiganConfig, params: ParamDict) -> Dict[str, np.ndarray]:
    out: Dict[str, np.ndarray] = {}
    for spec in conv_specs(cfg):
        out[spec.torch_prefix + ".weight"] = haiku_weight_to_torch(spec, params[spec.key]["w"])
        out[spec.torch_prefix + ".bias"] = np.ascontiguousarray(params[spec.key]["b"])
    return out


def _to_numpy(x):
    if hasattr(x, "detach"):
        return x.detach().cpu().numpy()
    return np.asarray(x)


# ---------------------------------------------------------------------------
# pickle I/O (mel2wave.py:35-36, convert_torch_model_to_haiku.py:60-62)
# ---------------------------------------------------------------------------
def check_params(cfg: HifiganConfig, params: Mapping) -> None:
    """Raise ValueError if ``params`` is not a complete Haiku generator dict for ``cfg``."""
    for spec in conv_specs(cfg):
        if spec.key not in params:
            raise ValueError(f"missing module {spec.key!r} in parameter dict")
        mod = params[spec.key]
        w, b = np.asarray(mod["w"]), np.asarray(mod["b"])
        if tuple(w.shape) != tuple(spec.w_shape):
            raise ValueError(f"{spec.key}: w has shape {w.shape}, expected {spec.w_shape}")
        if tuple(b.shape) != (spec.cout,):
            raise ValueError(f"{spec.key}: b has shape {b.shape}, expected {(spec.cout,)}")


def load_haiku_pickle(path) -> ParamDict:
    """Read ``hk_hifi.pickle``.  The file is a plain dict of numpy arrays (no JAX/Haiku
    classes inside), so the stock unpickler suffices."""
    with open(path, "rb") as f:
        raw = pickle.load(f)
    out: ParamDict = {}
    for k, mod in raw.items():
        out[str(k)] = {n: np.ascontiguousarray(np.asarray(a, dtype=np.float32)) for n, a in mod.items()}
    return out


def save_haiku_pickle(path, params: ParamDict) -> None:
    with open(path, "wb") as f:
        pickle.dump({k: {n: np.asarray(a) for n, a in mod.items()} for k, mod in params.items()}, f)


def iter_params(cfg: HifiganConfig, params: ParamDict) -> Iterator:
    for spec in conv_specs(cfg):
        yield spec, params[spec.key]["w"], params[spec.key]["b"]


def num_parameters(cfg: HifiganConfig) -> int:
    n = 0
    for s in conv_specs(cfg):
        n += s.k * s.cin * s.cout + s.cout
    return n
