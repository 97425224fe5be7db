"""The C-ABI library loads without a GPU and exports every symbol include/vtts_hifigan.h declares;
host-side planning (no compute) behaves as the header says."""
import ctypes as C
import re
from pathlib import Path

import pytest

from viettts_amd import _lib
from viettts_amd.hifigan.config import TINY, V1
from viettts_amd.hifigan.weights import conv_specs

REPO = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    from viettts_amd.csrc.build import build

    build()  # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def test_header_symbols_all_exported(lib):
    header = (REPO / "include" / "vtts_hifigan.h").read_text()
    declared = set(re.findall(r"\b(vtts_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.vtts_abi_version() == _lib.ABI_VERSION


def _create(lib, cfg, dtype=_lib.VTTS_F32):
    h = C.c_void_p(0)
    cs = _lib.make_cfg(cfg)
    _lib.check(lib, lib.vtts_hifigan_create(C.byref(cs), 0, dtype, C.byref(h)))
    return h


def test_plan_matches_python_inventory(lib):
    h = _create(lib, V1)
    n = C.c_int(0)
    _lib.check(lib, lib.vtts_hifigan_num_params(h, C.byref(n)))
    assert n.value == 156
    specs = conv_specs(V1)
    for i in range(n.value):
        key, which = C.c_char_p(), C.c_char_p()
        shape = (C.c_int64 * 3)()
        nd = C.c_int(0)
        _lib.check(lib, lib.vtts_hifigan_param_info(h, i, C.byref(key), C.byref(which), shape, C.byref(nd)))
        s = specs[i // 2]
        assert key.value.decode() == s.key
        if i % 2 == 0:
            assert which.value == b"w" and tuple(shape[:3]) == tuple(s.w_shape)
        else:
            assert which.value == b"b" and shape[0] == s.cout and nd.value == 1
    nb = C.c_size_t(0)
    _lib.check(lib, lib.vtts_hifigan_packed_bytes(h, C.byref(nb)))
    assert nb.value >= 13926017 * 4
    ws = C.c_size_t(0)
    _lib.check(lib, lib.vtts_hifigan_set_option(h, b"microbatch", 1))
    _lib.check(lib, lib.vtts_hifigan_workspace_bytes(h, 64, 1024, C.byref(ws)))
    assert ws.value == 2 * 4 * 8192 * 1024 * 4  # fp32 default: two micro-batches in flight, each four [C][L] buffers of the widest stage, one utterance
    _lib.check(lib, lib.vtts_hifigan_set_option(h, b"streams", 1))
    _lib.check(lib, lib.vtts_hifigan_workspace_bytes(h, 64, 1024, C.byref(ws)))
    assert ws.value == 4 * 8192 * 1024 * 4
    _lib.check(lib, lib.vtts_hifigan_set_option(h, b"streams", 0))
    hop = C.c_int64(0)
    _lib.check(lib, lib.vtts_hifigan_get_option(h, b"hop", C.byref(hop)))
    assert hop.value == 256
    lib.vtts_hifigan_destroy(h)


def test_error_paths(lib):
    h = _create(lib, TINY)
    # forward before weights are bound
    rc = lib.vtts_hifigan_forward(h, C.c_void_p(256), 1, 4, C.c_void_p(256), C.c_void_p(256), 1 << 30, None)
    assert rc == -2 and b"before pack" in lib.vtts_last_error()
    # unknown module / wrong shape
    buf = (C.c_float * 8)()
    shp = (C.c_int64 * 3)(1, 2, 3)
    assert lib.vtts_hifigan_set_param(h, b"generator/~/nope", b"w", buf, shp, 3) == -1
    assert lib.vtts_hifigan_set_param(h, b"generator/~/ups_0", b"w", buf, shp, 3) == -6
    assert lib.vtts_hifigan_set_param(h, b"generator/~/ups_0", b"q", buf, shp, 3) == -1
    # pack with parameters missing
    assert lib.vtts_hifigan_pack(h, C.c_void_p(256), 1 << 30, None) == -3
    assert b"never set" in lib.vtts_last_error()
    assert lib.vtts_hifigan_set_option(h, b"bogus", 1) == -1
    n = C.c_size_t(0)
    assert lib.vtts_hifigan_workspace_bytes(h, 0, 4, C.byref(n)) == -1
    # an utterance whose activations would overflow the kernels' 32-bit row * channel indexing is refused, not mis-indexed
    big = (1 << 31) // (TINY.upsample_initial_channel // 2 * TINY.upsample_rates[0]) + 1
    assert lib.vtts_hifigan_workspace_bytes(h, 1, big, C.byref(n)) == -1 and b"too long" in lib.vtts_last_error()
    lib.vtts_hifigan_destroy(h)
    # unsupported configurations are rejected at create()
    bad = _lib.make_cfg(V1)
    bad.resblock_kernel_sizes[0] = 4
    hh = C.c_void_p(0)
    assert lib.vtts_hifigan_create(C.byref(bad), 0, _lib.VTTS_F32, C.byref(hh)) == -1
    with pytest.raises(_lib.VttsError):
        _lib.check(lib, lib.vtts_hifigan_create(C.byref(bad), 0, _lib.VTTS_F32, C.byref(hh)))


def test_product_path_has_no_cpu_fallback():
    """Generator refuses CPU devices; mel2wave refuses to run without a GPU; nothing under
    viettts_amd/ imports the oracle."""
    import torch

    from viettts_amd.hifigan.generator import Generator

    with pytest.raises(ValueError):
        Generator(V1, device="cpu")
    if not torch.cuda.is_available():
        from viettts_amd.hifigan import mel2wave as m2w

        with pytest.raises((RuntimeError, FileNotFoundError)):
            m2w.mel2wave([[[0.0] * 80]])
    for p in (REPO / "viettts_amd").rglob("*.py"):
        src = p.read_text()
        assert "import oracle" not in src and "from oracle" not in src, p


def test_integration_md_stub_matches_the_abi(lib):
    """INTEGRATION.md §3 (the binding a maintainer of vietTTS/hifigan/mel2wave.py:20-41 would add): its struct is the header's
    struct — same fields, order, types, size — and every entry point it calls is exported.  (Round 2's document was one
    `int32_t resblock` short of ABI v2.)  The block is EXECUTED on the GPU by tests/test_gpu_parity.py."""
    import ast

    from _integration_stub import stub_source

    src = stub_source()
    tree = ast.parse(src)  # compiles
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "_Cfg"][0]
    ns = {"C": C}
    exec(compile(ast.Module([cls], []), "INTEGRATION.md", "exec"), ns)
    doc = ns["_Cfg"]
    assert [f[0] for f in doc._fields_] == [f[0] for f in _lib.CfgStruct._fields_]
    assert C.sizeof(doc) == C.sizeof(_lib.CfgStruct)
    for (_, a), (_, b) in zip(doc._fields_, _lib.CfgStruct._fields_):
        assert C.sizeof(a) == C.sizeof(b)
    # the header's own field list, parsed
    header = (REPO / "include" / "vtts_hifigan.h").read_text()
    body = header[header.index("typedef struct vtts_hifigan_cfg {") : header.index("} vtts_hifigan_cfg;")]
    assert re.findall(r"int32_t\s+([a-z_]+)", body) == [f[0] for f in doc._fields_]
    for sym in set(re.findall(r"_lib\.(vtts_[a-z_0-9]+)", src)):
        assert hasattr(lib, sym), sym


def test_built_library_has_no_packed_f32_valu():
    """No kernel of the library may contain v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: on MI355X their results are wrong while another wave
    of the SIMD streams bf16 MFMAs (round 4: the NAT decoder beside the bf16 generator; tools/kbench/pkfma_hazard.hip,
    profiles/r04_a_pkfma_findings.md), and they cannot co-issue with MFMAs anyway (profiles/r03_a_coissue_findings.md).  Every device file
    is built with -fno-slp-vectorize (viettts_amd/csrc/build.py); this disassembles what was actually built."""
    import sys
    from pathlib import Path

    repo = Path(__file__).resolve().parents[1]
    sys.path.insert(0, str(repo / "tools"))
    from check_no_packed_f32 import count_packed_f32

    from viettts_amd import _lib
    from viettts_amd.csrc import build

    assert all("-fno-slp-vectorize" in build.FILE_FLAGS.get(src, []) for src in build.SOURCES)
    counts = count_packed_f32(Path(_lib.default_lib_path()))
    assert len(counts) == len(build.SOURCES) and sum(m for _, m in counts.values()) > 5000  # every translation unit's code object was found and disassembled
    assert all(pk == 0 for pk, _ in counts.values()), counts


def test_experiment_includes_still_compile(tmp_path):
    """Development-only code lives OUT of the product sources (round 6: `tools/kbench/experiments/nat_*.inc`, included by `nat.hip` only under
    `-DVTTS_NAT_PERSIST` / `-DVTTS_NAT_PKFMA` / `-DVTTS_NAT_PP_EXP`; the timeline stamps of the stage / whole-ResBlock kernels under `-DVTTS_TIMELINE`).
    The findings under profiles/ cite those builds as provenance, so they must not rot: the three files that carry them compile with every switch on."""
    import subprocess

    from viettts_amd.csrc import build

    defs = ["-DVTTS_NAT_PERSIST=1", "-DVTTS_NAT_PKFMA=7", "-DVTTS_NAT_PP_EXP=1", "-DVTTS_TIMELINE=1"]
    procs = []
    for src in ("nat.hip", "kernels_bf16_stage.hip", "kernels_x3_rb.hip"):
        cmd = [build._hipcc(), *build.FLAGS, *build.FILE_FLAGS[src], *defs, "-c", str(build.CSRC / src), "-o", str(tmp_path / (src + ".o"))]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    for src, p in procs:
        _, err = p.communicate(timeout=900)
        assert p.returncode == 0, (src, err[-2000:])
