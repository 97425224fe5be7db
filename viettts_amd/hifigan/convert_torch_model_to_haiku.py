"""Drop-in for ``python -m vietTTS.hifigan.convert_torch_model_to_haiku`` (vietTTS/hifigan/convert_torch_model_to_haiku.py:65-79,
used by scripts/quick_start.sh:7): an upstream HiFi-GAN generator checkpoint ``g_XXXXXXXX`` -> ``hk_hifi.pickle`` under
``FLAGS.ckpt_dir``, the file ``mel2wave`` reads.

    python -m vietTTS.hifigan.convert_torch_model_to_haiku --config-file=assets/hifigan/config.json \\
        --checkpoint-file=assets/infore/hifigan/g_01140000

Same flags, same output path, same dict (module names, array layouts, float32 values): the weight-norm fold uses the
operation ``remove_weight_norm`` itself applies (``torch._weight_norm(v, g, 0)``, :31), the layout map is
viettts_amd.hifigan.weights.state_dict_to_haiku (:33-58).  Checked bit-for-bit against the reference converter's own
output on a weight-norm checkpoint (tests/golden/convert_tiny.npz, minted by oracle/make_golden.py).  The reference
instantiates its torch ``Generator`` to do this; here the state dict is mapped directly (no model code needed).
"""
from __future__ import annotations

import argparse
import json
import os

import numpy as np

from .config import FLAGS, HifiganConfig
from .weights import conv_specs, save_haiku_pickle, state_dict_to_haiku


def load_checkpoint(filepath, device="cpu"):
    """convert_torch_model_to_haiku.py:19-24."""
    import torch

    assert os.path.isfile(filepath)
    print("Loading '{}'".format(filepath))
    checkpoint_dict = torch.load(filepath, map_location=device)
    print("Complete.")
    return checkpoint_dict


def fold_weight_norm(cfg: HifiganConfig, state) -> dict:
    """``<prefix>.weight_g`` / ``weight_v`` -> ``<prefix>.weight`` exactly as ``remove_weight_norm()`` computes it
    (torch.nn.utils.weight_norm.WeightNorm.compute_weight = ``torch._weight_norm(v, g, dim=0)``); already-folded
    entries pass through."""
    import torch

    out = {}
    for spec in conv_specs(cfg):
        p = spec.torch_prefix
        if p + ".weight" in state:
            w = torch.as_tensor(state[p + ".weight"])
        elif p + ".weight_v" in state:
            w = torch._weight_norm(torch.as_tensor(state[p + ".weight_v"]).float(), torch.as_tensor(state[p + ".weight_g"]).float(), 0)
        else:
            raise KeyError(f"generator state dict has neither {p}.weight nor {p}.weight_v")
        out[p + ".weight"] = w.detach().cpu().numpy().astype(np.float32)
        out[p + ".bias"] = torch.as_tensor(state[p + ".bias"]).detach().cpu().numpy().astype(np.float32)
    return out


def convert_to_haiku(a, h, device="cpu"):
    """convert_torch_model_to_haiku.py:27-62: ``a.checkpoint_file`` -> ``FLAGS.ckpt_dir / "hk_hifi.pickle"``.
    ``h``: the JSON config as a mapping (or a HifiganConfig)."""
    cfg = h if isinstance(h, HifiganConfig) else HifiganConfig.from_dict(dict(h))
    state_dict_g = load_checkpoint(a.checkpoint_file, device)
    params = state_dict_to_haiku(cfg, fold_weight_norm(cfg, state_dict_g["generator"]))
    FLAGS.ckpt_dir.mkdir(parents=True, exist_ok=True)
    save_haiku_pickle(FLAGS.ckpt_dir / "hk_hifi.pickle", params)
    return params


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--checkpoint-file", required=True)
    parser.add_argument("--config-file", required=True)
    a = parser.parse_args(argv)
    with open(a.config_file) as f:
        h = json.loads(f.read())
    convert_to_haiku(a, h)


if __name__ == "__main__":
    main()
