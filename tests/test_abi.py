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
    assert l.cris_abi_version() == 1


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
