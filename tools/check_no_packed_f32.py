#!/usr/bin/env python3
"""Disassemble every gfx950 code object embedded in a built library and count packed-f32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 /
v_pk_add_f32).  None may be there: beside a wave that streams bf16 MFMAs their results are wrong on MI355X (viettts_amd/csrc/build.py,
profiles/r04_a_pkfma_findings.md).

    python tools/check_no_packed_f32.py [viettts_amd/lib/libvtts_hifigan.so]      -> prints {kernel file index: count}, exit 1 if any
"""
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
LLVM = Path("/opt/rocm/lib/llvm/bin")


def code_objects(lib: Path):
    """Yield (triple, bytes) of every device code object of every offload bundle in the file."""
    data = lib.read_bytes()
    pos = data.find(MAGIC)
    while pos >= 0:
        (n,) = struct.unpack_from("<Q", data, pos + len(MAGIC))
        off = pos + len(MAGIC) + 8
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24 : off + 24 + tl].decode()
            off += 24 + tl
            if "amdgcn" in triple and size:
                yield triple, data[pos + o : pos + o + size]
        pos = data.find(MAGIC, pos + 1)


def count_packed_f32(lib: Path):
    counts = {}
    with tempfile.TemporaryDirectory() as td:
        for i, (triple, blob) in enumerate(code_objects(lib)):
            f = Path(td) / f"dev{i}.co"
            f.write_bytes(blob)
            dis = subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(f)], capture_output=True, text=True).stdout
            counts[f"{i}:{triple}"] = (len(re.findall(r"\bv_pk_(?:fma|mul|add)_f32\b", dis)), len(re.findall(r"\bv_mfma_", dis)))
    return counts


if __name__ == "__main__":
    lib = Path(sys.argv[1] if len(sys.argv) > 1 else Path(__file__).resolve().parents[1] / "viettts_amd" / "lib" / "libvtts_hifigan.so")
    c = count_packed_f32(lib)
    for k, (pk, mf) in c.items():
        print(f"{k}: {pk} packed-f32 VALU instructions, {mf} MFMA instructions")
    sys.exit(1 if (not c or any(pk for pk, _ in c.values())) else 0)
