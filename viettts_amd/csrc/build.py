"""Build libvtts_hifigan.so (gfx950) in-tree with hipcc.

    python -m viettts_amd.csrc.build [--force] [--verbose]

The shared object lands in viettts_amd/lib/ so that it travels with the source tree to the GPU
box (it is git-ignored, not gpurun-ignored).  hipcc cross-compiles gfx950 without a GPU.
"""
from __future__ import annotations

import argparse
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent
ROOT = CSRC.parents[1]
LIBDIR = CSRC.parent / "lib"
OBJDIR = CSRC / "build"
SOURCES = ["engine.hip", "kernels_generic.hip", "kernels_f32_mfma.hip", "kernels_f32_pair.hip", "kernels_x3.hip", "kernels_x3_rb.hip", "kernels_bf16.hip", "kernels_bf16_rbg.hip", "kernels_bf16_rbk.hip", "kernels_bf16_stage.hip", "kernels_bf16_up.hip", "nat.hip"]
HEADERS = ["vtts_internal.h", "device_common.h", "bf16_common.h", str(ROOT / "include" / "vtts_hifigan.h"), str(ROOT / "include" / "vtts_nat.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# No kernel of this library may contain packed-f32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32); hipcc's SLP vectoriser is
# what forms them, so every device file is built with -fno-slp-vectorize.  Two reasons, both measured on MI355X:
#  * performance (round 3): they cannot issue while the SIMD's other wave streams MFMAs (bf16_common.h, profiles/r03_a_coissue_findings.md);
#  * CORRECTNESS (round 4): a wave's v_pk_fma_f32 results come out WRONG (the low halves of the packed pairs) while another wave of the same
#    SIMD streams v_mfma_f32_32x32x16_bf16 — the NAT decoder beside the bf16 generator, i.e. the overlapped text -> waveform pipeline,
#    computed wrong mel frames for its even sentences (tools/experiments/r04/diag_pipe3.py; isolated in tools/kbench/pkfma_hazard.hip;
#    profiles/r04_a_pkfma_findings.md).  Kernels that never run beside the bf16 engine would be safe with them, but a caller may put any
#    two handles on two streams, so none keeps them (the fp32 convolutions lose ~1 %).
# (The LOOP vectoriser can form them too, from a lane-strided scalar loop: such loops carry `#pragma clang loop vectorize(disable)` —
# nat.hip: nat_gates_mix_k — and tests/test_cabi.py disassembles the built library, which is what holds the rule.)
FILE_FLAGS = {name: ["-fno-slp-vectorize"] for name in SOURCES}
LIBNAME = "libvtts_hifigan.so"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest() -> str:
    h = hashlib.sha256()
    for name in SOURCES + HEADERS + [__file__]:
        p = Path(name) if os.path.isabs(str(name)) else CSRC / name
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    return h.hexdigest()


def lib_path() -> Path:
    return LIBDIR / LIBNAME


def build(force: bool = False, verbose: bool = False, extra_flags=(), libname: str = LIBNAME) -> Path:
    """extra_flags/libname build an experiment variant next to the product library (A/B runs)."""
    global FLAGS
    LIBDIR.mkdir(parents=True, exist_ok=True)
    OBJDIR.mkdir(parents=True, exist_ok=True)
    stamp = LIBDIR / (libname + ".sha256")
    base_flags = list(FLAGS)
    FLAGS = base_flags + list(extra_flags)
    try:
        return _build(force, verbose, stamp, LIBDIR / libname, libname)
    finally:
        FLAGS = base_flags


def _build(force, verbose, stamp, out, libname) -> Path:
    dig = _digest()
    if not force and out.exists() and stamp.exists() and stamp.read_text().strip() == dig:
        return out
    hipcc = _hipcc()

    def compile_one(src: str) -> Path:
        obj = OBJDIR / (Path(src).stem + "." + libname + ".o")
        per_file = [] if os.environ.get("VTTS_BUILD_NO_FILE_FLAGS") else FILE_FLAGS.get(src, [])  # experiment builds only (A/B of the flag itself)
        if src in os.environ.get("VTTS_BUILD_NOSLP_FILES", "").split(","):
            per_file = ["-fno-slp-vectorize"]
        cmd = [hipcc, *FLAGS, *per_file, "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # Link WITHOUT a DT_NEEDED on libamdhip64: PyTorch-ROCm bundles its own HIP runtime
    # (torch/lib/libamdhip64.so, a different soname from /opt/rocm's libamdhip64.so.7), and a process
    # must not end up with two runtimes.  The hip* symbols stay undefined here and bind at dlopen to
    # the runtime the host already loaded (viettts_amd/_lib.py promotes it to the global scope; a C
    # host links -lamdhip64 itself).
    cxx = os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib", "llvm", "bin", "clang++")
    if not os.path.exists(cxx):
        cxx = "/opt/rocm/lib/llvm/bin/clang++"
    cmd = [cxx, "-shared", "-fPIC", "-o", str(out), *map(str, objs)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    stamp.write_text(dig + "\n")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--define", action="append", default=[], help="extra -D flags for an experiment variant")
    ap.add_argument("--libname", default=LIBNAME)
    a = ap.parse_args()
    print(build(a.force, a.verbose, ["-D" + d for d in a.define], a.libname))
