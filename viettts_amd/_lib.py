"""ctypes binding of include/vtts_hifigan.h.

The product path has NO CPU fallback: if the shared library is missing or a call fails, a
:class:`VttsError` (or ``OSError`` from the loader) propagates.  Build the library with
``python -m viettts_amd.csrc.build`` (``__graft_entry__.build()`` does).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

ABI_VERSION = 2
MAX_UPSAMPLES = 8
MAX_KERNELS = 4

VTTS_F32 = 0
VTTS_BF16 = 1
VTTS_BF16X3 = 2

STATUS_NAMES = {
    0: "VTTS_OK",
    -1: "VTTS_ERR_INVALID",
    -2: "VTTS_ERR_STATE",
    -3: "VTTS_ERR_MISSING",
    -4: "VTTS_ERR_HIP",
    -5: "VTTS_ERR_NOMEM",
    -6: "VTTS_ERR_SHAPE",
}

# Every symbol include/vtts_hifigan.h declares (tests/test_cabi.py checks the header against this).
EXPORTS = (
    "vtts_abi_version",
    "vtts_last_error",
    "vtts_hifigan_create",
    "vtts_hifigan_destroy",
    "vtts_hifigan_set_param",
    "vtts_hifigan_num_params",
    "vtts_hifigan_param_info",
    "vtts_hifigan_packed_bytes",
    "vtts_hifigan_pack",
    "vtts_hifigan_bind_packed",
    "vtts_hifigan_workspace_bytes",
    "vtts_hifigan_forward",
    "vtts_hifigan_forward_ragged",
    "vtts_hifigan_tap_elems",
    "vtts_hifigan_forward_tap",
    "vtts_hifigan_run_module",
    "vtts_hifigan_run_pair",
    "vtts_hifigan_set_option",
    "vtts_hifigan_get_option",
    "vtts_hifigan_profile_read",
    "vtts_hifigan_profile_kernel",
)


# Every symbol include/vtts_nat.h declares.
NAT_EXPORTS = (
    "vtts_nat_duration_create",
    "vtts_nat_duration_destroy",
    "vtts_nat_duration_set_param",
    "vtts_nat_duration_num_params",
    "vtts_nat_duration_param_info",
    "vtts_nat_duration_packed_bytes",
    "vtts_nat_duration_pack",
    "vtts_nat_duration_bind_packed",
    "vtts_nat_duration_workspace_bytes",
    "vtts_nat_duration_forward",
    "vtts_nat_acoustic_create",
    "vtts_nat_acoustic_destroy",
    "vtts_nat_acoustic_set_param",
    "vtts_nat_acoustic_num_params",
    "vtts_nat_acoustic_param_info",
    "vtts_nat_acoustic_packed_bytes",
    "vtts_nat_acoustic_pack",
    "vtts_nat_acoustic_bind_packed",
    "vtts_nat_acoustic_workspace_bytes",
    "vtts_nat_acoustic_set_option",
    "vtts_nat_acoustic_get_option",
    "vtts_nat_acoustic_keep_masks",
    "vtts_nat_acoustic_keep_masks_haiku",
    "vtts_nat_acoustic_keep_masks_haiku_mode",
    "vtts_nat_acoustic_forward",
    "vtts_nat_acoustic_forward_groups",
    "vtts_nat_acoustic_wait_group",
    "vtts_nat_acoustic_encode",
    "vtts_nat_acoustic_forward_from_encoder",
)


class NatDurationCfg(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("lstm_dim", C.c_int32)]


class NatAcousticCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vocab_size", "encoder_dim", "decoder_dim", "prenet_dim", "mel_dim", "postnet_dim")]


class VttsError(RuntimeError):
    """A C-ABI call returned a negative vtts_status."""

    def __init__(self, status: int, message: str):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")


class CfgStruct(C.Structure):
    _fields_ = [
        ("num_mels", C.c_int32),
        ("upsample_initial_channel", C.c_int32),
        ("num_upsamples", C.c_int32),
        ("upsample_rates", C.c_int32 * MAX_UPSAMPLES),
        ("upsample_kernel_sizes", C.c_int32 * MAX_UPSAMPLES),
        ("num_kernels", C.c_int32),
        ("resblock_kernel_sizes", C.c_int32 * MAX_KERNELS),
        ("resblock_dilation_sizes", (C.c_int32 * 3) * MAX_KERNELS),
        ("resblock", C.c_int32),
    ]


def default_lib_path() -> Path:
    env = os.environ.get("VTTS_HIFIGAN_LIB")
    if env:
        return Path(env)
    return Path(__file__).resolve().parent / "lib" / "libvtts_hifigan.so"


_LIB: Optional[C.CDLL] = None
_HIP_RT = None


def _load_hip_runtime():
    """libvtts_hifigan.so carries no DT_NEEDED for the HIP runtime (csrc/build.py): bind it to the
    ONE runtime the process uses.  With PyTorch-ROCm that is torch's bundled libamdhip64.so — streams
    and device pointers handed across the C ABI come from it — otherwise the system one."""
    global _HIP_RT
    if _HIP_RT is not None:
        return _HIP_RT
    cands = []
    try:
        import torch  # noqa: F401  (loads its runtime first)

        cands.append(Path(torch.__file__).resolve().parent / "lib" / "libamdhip64.so")
    except Exception:
        pass
    cands += [Path("/opt/rocm/lib/libamdhip64.so")]
    err = None
    for c in cands:
        if c.exists():
            try:
                _HIP_RT = C.CDLL(str(c), mode=C.RTLD_GLOBAL)
                return _HIP_RT
            except OSError as e:  # pragma: no cover
                err = e
    raise OSError(f"no HIP runtime (libamdhip64) could be loaded: {err}")



def load(path=None) -> C.CDLL:
    """dlopen the HIP extension and declare prototypes.  Raises OSError if it is not built."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = Path(path) if path else default_lib_path()
    if not p.exists():
        raise OSError(
            f"HIP extension {p} not found — build it with `python -m viettts_amd.csrc.build` "
            "(there is no CPU fallback on the product path)"
        )
    _load_hip_runtime()
    lib = C.CDLL(str(p))
    vp, cp, sz, i64 = C.c_void_p, C.c_char_p, C.c_size_t, C.c_int64
    fp = C.POINTER(C.c_float)
    sigs = {
        "vtts_abi_version": (C.c_int, []),
        "vtts_last_error": (cp, []),
        "vtts_hifigan_create": (C.c_int, [C.POINTER(CfgStruct), C.c_int, C.c_int, C.POINTER(vp)]),
        "vtts_hifigan_destroy": (None, [vp]),
        "vtts_hifigan_set_param": (C.c_int, [vp, cp, cp, vp, C.POINTER(i64), C.c_int]),
        "vtts_hifigan_num_params": (C.c_int, [vp, C.POINTER(C.c_int)]),
        "vtts_hifigan_param_info": (C.c_int, [vp, C.c_int, C.POINTER(cp), C.POINTER(cp), C.POINTER(i64), C.POINTER(C.c_int)]),
        "vtts_hifigan_packed_bytes": (C.c_int, [vp, C.POINTER(sz)]),
        "vtts_hifigan_pack": (C.c_int, [vp, vp, sz, vp]),
        "vtts_hifigan_bind_packed": (C.c_int, [vp, vp, sz]),
        "vtts_hifigan_workspace_bytes": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(sz)]),
        "vtts_hifigan_forward": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, sz, vp]),
        "vtts_hifigan_forward_ragged": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp, vp, sz, vp]),
        "vtts_hifigan_tap_elems": (C.c_int, [vp, cp, C.c_int, C.c_int, C.POINTER(sz)]),
        "vtts_hifigan_forward_tap": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, sz, vp, cp, vp]),
        "vtts_hifigan_run_module": (C.c_int, [vp, cp, vp, C.c_int, C.c_int, C.c_float, vp, vp, vp]),
        "vtts_hifigan_run_pair": (C.c_int, [vp, cp, vp, C.c_int, C.c_int, vp, vp]),
        "vtts_hifigan_set_option": (C.c_int, [vp, cp, i64]),
        "vtts_hifigan_get_option": (C.c_int, [vp, cp, C.POINTER(i64)]),
        "vtts_hifigan_profile_read": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double), C.c_int]),
        "vtts_hifigan_profile_kernel": (cp, [vp]),
        "vtts_nat_duration_create": (C.c_int, [C.POINTER(NatDurationCfg), C.c_int, C.POINTER(vp)]),
        "vtts_nat_duration_destroy": (None, [vp]),
        "vtts_nat_duration_set_param": (C.c_int, [vp, cp, cp, vp, C.POINTER(i64), C.c_int]),
        "vtts_nat_duration_num_params": (C.c_int, [vp, C.POINTER(C.c_int)]),
        "vtts_nat_duration_param_info": (C.c_int, [vp, C.c_int, C.POINTER(cp), C.POINTER(cp), C.POINTER(i64), C.POINTER(C.c_int)]),
        "vtts_nat_duration_packed_bytes": (C.c_int, [vp, C.POINTER(sz)]),
        "vtts_nat_duration_pack": (C.c_int, [vp, vp, sz, vp]),
        "vtts_nat_duration_bind_packed": (C.c_int, [vp, vp, sz]),
        "vtts_nat_duration_workspace_bytes": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(sz)]),
        "vtts_nat_duration_forward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp, vp, sz, vp]),
        "vtts_nat_acoustic_create": (C.c_int, [C.POINTER(NatAcousticCfg), C.c_int, C.POINTER(vp)]),
        "vtts_nat_acoustic_destroy": (None, [vp]),
        "vtts_nat_acoustic_set_param": (C.c_int, [vp, cp, cp, vp, C.POINTER(i64), C.c_int]),
        "vtts_nat_acoustic_num_params": (C.c_int, [vp, C.POINTER(C.c_int)]),
        "vtts_nat_acoustic_param_info": (C.c_int, [vp, C.c_int, C.POINTER(cp), C.POINTER(cp), C.POINTER(i64), C.POINTER(C.c_int)]),
        "vtts_nat_acoustic_packed_bytes": (C.c_int, [vp, C.POINTER(sz)]),
        "vtts_nat_acoustic_pack": (C.c_int, [vp, vp, sz, vp]),
        "vtts_nat_acoustic_bind_packed": (C.c_int, [vp, vp, sz]),
        "vtts_nat_acoustic_workspace_bytes": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(sz)]),
        "vtts_nat_acoustic_set_option": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "vtts_nat_acoustic_get_option": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_int)]),
        "vtts_nat_acoustic_keep_masks": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp]),
        "vtts_nat_acoustic_keep_masks_haiku": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, vp, vp]),
        "vtts_nat_acoustic_keep_masks_haiku_mode": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, vp, vp]),
        "vtts_nat_acoustic_forward": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, sz, vp]),
        "vtts_nat_acoustic_forward_groups": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, sz, vp, C.c_int,
                                                       C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "vtts_nat_acoustic_wait_group": (C.c_int, [vp, C.c_int, vp]),
        "vtts_nat_acoustic_encode": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp, vp, sz, vp]),
        "vtts_nat_acoustic_forward_from_encoder": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, sz, vp, C.c_int,
                                                             C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.vtts_abi_version()
    if got != ABI_VERSION:
        raise OSError(f"{p}: ABI version {got}, this binding expects {ABI_VERSION}")
    if path is None:
        _LIB = lib
    return lib


def check(lib: C.CDLL, status: int) -> None:
    if status != 0:
        msg = lib.vtts_last_error()
        raise VttsError(status, msg.decode() if msg else "")


def make_cfg(cfg) -> CfgStruct:
    """viettts_amd.hifigan.config.HifiganConfig -> vtts_hifigan_cfg."""
    cfg.validate()
    if cfg.num_upsamples > MAX_UPSAMPLES or cfg.num_kernels > MAX_KERNELS:
        raise ValueError("architecture exceeds the C ABI's fixed array sizes")
    s = CfgStruct()
    s.num_mels = cfg.num_mels
    s.upsample_initial_channel = cfg.upsample_initial_channel
    s.num_upsamples = cfg.num_upsamples
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        s.upsample_rates[i] = int(u)
        s.upsample_kernel_sizes[i] = int(k)
    s.num_kernels = cfg.num_kernels
    for j, (k, d) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
        s.resblock_kernel_sizes[j] = int(k)
        for z in range(len(d)):
            s.resblock_dilation_sizes[j][z] = int(d[z])
    s.resblock = int(cfg.resblock)
    return s
