"""Generator hyper-parameters and checkpoint locations.

Mirrors what the reference reads at the boundary:
  * ``assets/hifigan/config.json`` relative to the CWD (vietTTS/hifigan/mel2wave.py:21-26);
    only the fields Generator.__init__ consumes matter (vietTTS/hifigan/model.py:81-106).
  * ``FLAGS.ckpt_dir = ./assets/infore/hifigan`` (vietTTS/hifigan/config.py:5-6) holding
    ``hk_hifi.pickle`` (vietTTS/hifigan/mel2wave.py:35).
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from pathlib import Path
from typing import Sequence


class FLAGS:
    """Same attribute the reference exposes (vietTTS/hifigan/config.py:5-6)."""

    ckpt_dir = Path("./assets/infore/hifigan")
    config_file = Path("assets/hifigan/config.json")
    dtype = None  # engine of the drop-in mel2wave: None / "f32" = the 1e-4-parity engine, "bf16" = the throughput engine


@dataclass(frozen=True)
class HifiganConfig:
    """The architecture-defining subset of the HiFi-GAN JSON config.

    Defaults are HiFi-GAN V1 as shipped by the reference
    (assets/hifigan/config.json:2,11-15,19,22,25).
    """

    resblock: str = "1"
    upsample_rates: Sequence[int] = (8, 8, 2, 2)
    upsample_kernel_sizes: Sequence[int] = (16, 16, 4, 4)
    upsample_initial_channel: int = 512
    resblock_kernel_sizes: Sequence[int] = (3, 7, 11)
    resblock_dilation_sizes: Sequence[Sequence[int]] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    num_mels: int = 80
    sampling_rate: int = 16000

    # ---- derived ---------------------------------------------------------
    @property
    def num_upsamples(self) -> int:
        return len(self.upsample_rates)

    @property
    def num_kernels(self) -> int:
        return len(self.resblock_kernel_sizes)

    @property
    def hop(self) -> int:
        """Output samples per mel frame (256 for V1)."""
        h = 1
        for u in self.upsample_rates:
            h *= int(u)
        return h

    def stage_channels(self, i: int) -> int:
        """Channels after upsample stage ``i`` (model.py:90, :98)."""
        return self.upsample_initial_channel // (2 ** (i + 1))

    def validate(self) -> None:
        if self.resblock not in ("1", "2"):
            raise ValueError("resblock must be '1' (ResBlock1, model.py:13-51) or '2' (ResBlock2, model.py:54-74), got %r" % (self.resblock,))
        if len(self.upsample_rates) != len(self.upsample_kernel_sizes):
            raise ValueError("upsample_rates and upsample_kernel_sizes differ in length")
        if len(self.resblock_kernel_sizes) != len(self.resblock_dilation_sizes):
            raise ValueError("resblock_kernel_sizes and resblock_dilation_sizes differ in length")
        for k in self.resblock_kernel_sizes:
            if k % 2 != 1:
                raise ValueError("resblock kernel sizes must be odd (length-preserving padding)")
        nd = 3 if self.resblock == "1" else 2
        for d in self.resblock_dilation_sizes:
            if len(d) != nd:
                raise ValueError("ResBlock%s takes exactly %d dilations per kernel size (model.py:%s)" % (self.resblock, nd, "15,43" if nd == 3 else "55,68"))
        if self.upsample_initial_channel % (2 ** self.num_upsamples) != 0:
            raise ValueError("upsample_initial_channel must be divisible by 2**num_upsamples")

    @staticmethod
    def from_dict(d: dict) -> "HifiganConfig":
        kw = {}
        for name in (
            "resblock",
            "upsample_rates",
            "upsample_kernel_sizes",
            "upsample_initial_channel",
            "resblock_kernel_sizes",
            "resblock_dilation_sizes",
            "num_mels",
            "sampling_rate",
        ):
            if name in d:
                v = d[name]
                if isinstance(v, list):
                    v = tuple(tuple(e) if isinstance(e, list) else e for e in v)
                kw[name] = v
        cfg = HifiganConfig(**kw)
        cfg.validate()
        return cfg

    @staticmethod
    def from_json(path) -> "HifiganConfig":
        with open(path) as f:
            return HifiganConfig.from_dict(json.load(f))


V1 = HifiganConfig()

# A deliberately tiny architecture with the same structure as V1 (4 upsample stages,
# 3x3 ResBlock1 grid, hop 256) used by fixtures whose weights are small enough to
# commit under tests/golden/.
TINY = HifiganConfig(upsample_initial_channel=32)

# The same, with ResBlock2 (vietTTS/hifigan/model.py:54-74: two convolutions per block, each with its own residual) and the
# kernel sizes / dilations of the upstream V3 config.  The reference builds this generator (model.py:86) but cannot LOAD a
# checkpoint for it: its converter names the modules res_block2_N/~/convs_Z (convert_torch_model_to_haiku.py:45-46) while the
# Haiku model creates res_block1_N/~/conv1_d[_1] (model.py:105, default hk.Conv1D names).  Here the parameter dict uses the
# names the Haiku MODEL creates, and the converter of this repo writes those.
TINY2 = HifiganConfig(resblock="2", upsample_initial_channel=32, resblock_kernel_sizes=(3, 5, 7),
                      resblock_dilation_sizes=((1, 2), (2, 6), (3, 12)))
