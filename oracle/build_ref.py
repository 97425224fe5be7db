"""ORACLE tooling — builds the REAL reference generator into a binary that can travel to the GPU box.

The reference's CPU path for the hot path is ``vietTTS/hifigan/torch_model.py::Generator`` (:156-218), the PyTorch
model the Haiku weights are converted from (the JAX/Haiku path cannot be installed offline).  Python
sources cannot travel (/root/reference does not exist on the GPU box, and reference sources are never copied into this
repo), so — exactly as a C reference would be compiled to ``oracle/_ref/*.so`` — the module is imported HERE from where
it lies, ``remove_weight_norm()`` applied (as the reference's converter does, convert_torch_model_to_haiku.py:32), and
compiled with ``torch.jit.trace`` into ``oracle/_ref/torch_generator_v1.pt.gz``: a TorchScript archive of the reference's
own graph of aten ops (conv1d / conv_transpose1d / leaky_relu / add / div / tanh), parameters as loadable state.  The
archive is written with ZEROED parameters and gzip-ed (56 MB -> ~60 KB: every gpurun call pushes the tree); its users load
the weights they want with ``load_state_dict`` (:func:`load_reference_archive`).
``oracle/_ref/`` is git-ignored (built artefact) but travels with gpurun like the HIP library.

Consumers: ``bench.py::cpu_baseline`` (kind "reference") and tests/test_oracle_golden.py (the archive against the
committed golden vectors).  Never the product path.

Usage:  python oracle/build_ref.py        (no-op with a message when /root/reference is absent)
"""
from __future__ import annotations

import io
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
OUT = REPO / "oracle" / "_ref"


ARCHIVE = OUT / "torch_generator_v1.pt.gz"


def load_reference_archive(state_dict=None):
    """The built reference generator (TorchScript, CPU), optionally with ``state_dict`` (torch tensors, the reference's own
    ``conv_pre.weight`` ... names) loaded; ``None`` if the archive is not there."""
    import gzip
    import io

    import torch

    if not ARCHIVE.exists():
        return None
    with gzip.open(ARCHIVE, "rb") as f:
        ts = torch.jit.load(io.BytesIO(f.read()), map_location="cpu").eval()
    if state_dict is not None:
        ts.load_state_dict(state_dict, strict=True)
    return ts


def main() -> int:
    if not (REF / "vietTTS/hifigan/torch_model.py").exists():
        print("oracle/build_ref.py: /root/reference not present — keeping whatever oracle/_ref/ holds")
        return 0
    import warnings

    import torch

    sys.path.insert(0, str(REPO))
    from oracle.make_golden import _import_reference, reference_generator  # noqa: E402  (the loader the golden vectors were minted with)

    tm, _ = _import_reference()  # the reference's vietTTS package, ahead of this repo's drop-in shim of the same name
    from viettts_amd.hifigan.config import V1  # noqa: E402
    from viettts_amd.hifigan.synth import synthetic_mel, synthetic_params  # noqa: E402

    OUT.mkdir(parents=True, exist_ok=True)
    params = synthetic_params(V1, 4321, "scaled")
    g = reference_generator(tm, V1, params, torch.float32)
    x = torch.from_numpy(synthetic_mel(1, 24, 5)).permute(0, 2, 1).contiguous()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.no_grad():
            ts = torch.jit.trace(g, x, check_trace=False)
    # the trace must not have baked the example's shape in: compare with the eager reference at another (B, T)
    x2 = torch.from_numpy(synthetic_mel(3, 41, 6)).permute(0, 2, 1).contiguous()
    with torch.no_grad():
        d = float((ts(x2) - g(x2)).abs().max())
    assert d == 0.0, f"traced reference differs from the eager reference at a new shape: {d}"
    import gzip
    import io

    nparam = int(sum(v.numel() for v in ts.state_dict().values()))
    with torch.no_grad():
        for v in ts.state_dict().values():
            v.zero_()  # users load their own weights; zeros make the archive ~60 KB
    bio = io.BytesIO()
    torch.jit.save(ts, bio)
    path = ARCHIVE
    with gzip.open(path, "wb", compresslevel=6) as f:
        f.write(bio.getvalue())
    old = OUT / "torch_generator_v1.pt"
    if old.exists():
        old.unlink()
    meta = {"source": "vietTTS/hifigan/torch_model.py:156-218 (Generator, weight norm removed), torch.jit.trace",
            "torch": torch.__version__, "state_dict_keys": len(ts.state_dict()),
            "params": nparam}
    (OUT / "torch_generator_v1.json").write_text(json.dumps(meta, indent=1))
    print(f"built {path} ({path.stat().st_size / 1e3:.0f} KB, {meta['params']} parameters)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
