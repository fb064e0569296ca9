"""CPU-only checks of the C ABI: the library loads, exports every symbol include/cris_hip.h declares,
and the ctypes mirrors in cris/pytorch_amd/hip.py match the C structs field by field (offsets from a
gcc-compiled probe that includes the real header).  No compute calls here (no GPU)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cris_hip.h")


@pytest.fixture(scope="module")
def lib():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    from cris.pytorch_amd import hip
    return hip, hip.load()


def test_exports_match_header(lib):
    hip, l = lib
    src = open(HEADER).read()
    declared = set(re.findall(r"\b(cris_[a-z0-9_]+)\s*\(", src))
    declared = {d for d in declared if not d.endswith("_params") and not d.endswith("_desc")}
    assert declared == set(hip.EXPORTS), (declared ^ set(hip.EXPORTS))
    for name in declared:
        assert getattr(l, name) is not None
    # one number in three places: the header, the library built from it, the ctypes binding (which refuses any other library)
    assert l.cris_abi_version() == hip.ABI_VERSION == int(re.search(r"#define CRIS_ABI_VERSION (\d+)", src).group(1))


def test_struct_layouts(lib):
    hip, l = lib
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % HEADER, 'int main(){']
    for cname, st in hip.STRUCTS.items():
        lines.append('printf("%s.__size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in st._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines.append('return 0;}')
    with tempfile.TemporaryDirectory() as d:
        cfile = os.path.join(d, "probe.c")
        open(cfile, "w").write("\n".join(lines))
        exe = os.path.join(d, "probe")
        subprocess.check_call(["gcc", "-o", exe, cfile])
        out = subprocess.check_output([exe]).decode().split("\n")
    got = dict(ln.split(" ") for ln in out if ln)
    for cname, st in hip.STRUCTS.items():
        assert int(got[cname + ".__size"]) == C.sizeof(st), cname
        assert l.cris_sizeof(cname.encode()) == C.sizeof(st), cname
        for fname, _ in st._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(st, fname).offset, (cname, fname)
    # number of fields must match too (a missing trailing field would not shift any offset)
    src = open(HEADER).read()
    for cname, st in hip.STRUCTS.items():
        body = re.search(r"typedef struct \{([^{}]*)\} %s;" % cname, src).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        nfields = 0
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                nfields += decl.count(",") + 1
        assert nfields == len(st._fields_), (cname, nfields, len(st._fields_))


def test_echo_roundtrip(lib):
    hip, l = lib
    p = hip.ConvGemmParams()
    p.A, p.outT, p.T_sec_stride = 0x1000, 0x2000, 123456789012
    p.lda, p.C, p.pad, p.ldb, p.K, p.act, p.out_f32, p.T_E = 7, 11, 13, 17, 19, 2, 1, 512
    p.drop_p, p.drop_seed, p.drop_stream = 0.5, 99, 5
    h = 0
    for v in (0x1000, 0x2000, 123456789012, 7, 11, 13, 17, 19, 2, 1, 512, 500, 99, 5):
        h = (h * 31 + v) & 0xFFFFFFFFFFFFFFFF
    got = l.cris_echo_conv_gemm(C.byref(p)) & 0xFFFFFFFFFFFFFFFF
    assert got == h


def test_argument_validation_without_gpu(lib):
    """Launchers validate geometry on the host before touching the device and report through
    cris_last_error()."""
    hip, l = lib
    p = hip.ConvGemmParams()
    rc = l.cris_conv_gemm(C.byref(p), None)
    assert rc != 0 and b"null operand" in l.cris_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    from cris.pytorch_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", "/nonexistent/libcris_hip.so")
    with pytest.raises(hip.HipLibraryError):
        hip.load()


def test_no_packed_fp32_valu_instructions():
    """csrc/build.py FLAGS: v_pk_{mul,add,fma}_f32 gave wrong results under two-stream concurrency on MI355X
    (profiles/r02_packed_fp32_concurrency.md) - the shipped code object must not contain any"""
    from cris.pytorch_amd.csrc import build
    lib = build.build()
    found = build.packed_fp32_ops(lib)
    if found is None:
        pytest.skip("llvm-objdump not available")
    assert found == {}, found


def test_comm_entry_points_resolve_rccl_and_fail_loudly_without_a_gpu(lib):
    """cris_comm_*: RCCL is resolved at run time (no link-time dependency: `ldd` must not list it); without a device the
    communicator cannot be created and says why."""
    import torch
    hip, l = lib
    deps = subprocess.check_output(["ldd", hip.LIB_PATH]).decode()
    assert "rccl" not in deps
    path = l.cris_comm_rccl_path()
    assert path, l.cris_last_error()
    buf = (C.c_ubyte * 256)()
    assert l.cris_comm_unique_id(buf) == 0, l.cris_last_error()
    assert any(buf[:128]) and any(buf[128:]) and bytes(buf[:128]) != bytes(buf[128:])     # two distinct communicator ids
    if not torch.cuda.is_available():
        h = C.c_void_p()
        assert l.cris_comm_init(0, 1, buf, C.byref(h)) != 0
        assert b"cris_comm_init" in l.cris_last_error()
    assert l.cris_comm_init(3, 2, buf, None) != 0                                         # bad arguments are rejected


def test_host_side_selection_logic_without_gpu(lib):
    """the launch-geometry decisions that are pure host arithmetic (no device call): tile-variant table, weight-gradient tile
    choice, BatchNorm partial-list rows, Adam block counts"""
    hip, l = lib
    l.cris_conv_gemm_variant_name.restype = C.c_char_p
    names = [l.cris_conv_gemm_variant_name(i).decode() for i in range(l.cris_conv_gemm_num_variants())]
    assert names == ["skinny1", "skinny9", "skinny9s", "128x64", "64x64", "64x128", "128x128", "8w256x256", "8w256x128", "8w128x256",
                     "8w128x128", "64x64k2"]
    # weight gradients: the 8-wave 256x256 tile for long, wide reductions only; params.tile forces either
    got = []
    for M, N, K, tile in [(21632, 512, 4608, 0), (21632, 256, 1152, 0), (5408, 512, 4608, 0), (21632, 128, 4608, 0),
                          (86528, 256, 4608, 0), (300, 200, 648, 256), (21632, 512, 4608, 128)]:
        p = hip.WgradParams()
        p.M, p.N, p.K, p.tile = M, N, K, tile
        got.append(l.cris_conv_wgrad_tile(C.byref(p)))
    assert got == [256, 128, 128, 128, 256, 256, 128], got
    # BatchNorm partial lists: up to 512 parts one launch, beyond that room for the 64 first-level slices
    assert [l.cris_bn_partials_rows(n) for n in (1, 85, 512, 513, 2704)] == [1, 85, 512, 577, 2768]
    # Adam: plain tensors in 8192-element blocks, 1-tap packed weights in 64x64 tiles, 9-tap ones in 32-row tiles
    be = l.cris_adam_block_elems()
    assert be == 8192
    d = hip.AdamDesc()
    d.n = 3 * be + 1
    assert l.cris_adam_blocks(C.byref(d)) == 4
    d = hip.AdamDesc()
    d.dstF, d.N, d.cin, d.taps = 0x1000, 130, 200, 0
    assert l.cris_adam_blocks(C.byref(d)) == 3 * 4
    d.taps = 9
    assert l.cris_adam_blocks(C.byref(d)) == 5 * 4
    # general epilogue / lean epilogue, tile choice of a mid-size and a large problem (names through the stat-row contract)
    p = hip.ConvGemmParams()
    p.M, p.N, p.K, p.C, p.KH, p.KW, p.stride, p.pad, p.H, p.W, p.OH, p.OW, p.Bn = 136, 512, 512, 512, 1, 1, 1, 0, 17, 1, 17, 1, 8
    assert l.cris_conv_gemm_stat_rows(C.byref(p)) == 32            # M <= 144, K < 1024: the 64x64 tile (32-row wave tiles)
    p.K = p.C = 2048
    assert l.cris_conv_gemm_stat_rows(C.byref(p)) == 16            # split-K skinny kernels
    assert l.cris_conv_gemm_ws_floats(C.byref(p), -1) > 0
