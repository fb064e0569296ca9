"""Build libcris_hip.so for gfx950 (hipcc cross-compiles without a GPU).  In-tree output:
cris/pytorch_amd/csrc/libcris_hip.so (git-ignored, shipped to the GPU box by gpurun)."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.hip", "gemm.hip", "gemm8.hip", "wgrad.hip", "norm.hip", "attention.hip", "elementwise.hip", "smallf32.hip", "evalpost.hip", "inputpipe.hip", "p2p.hip", "comm.hip", "jpeg.hip", "png.hip"]
LIB = os.path.join(HERE, "libcris_hip.so")
STAMP = os.path.join(HERE, ".build_stamp")
# -fno-slp-vectorize -fno-vectorize: keep packed-FP32 VALU instructions (v_pk_mul/add/fma_f32) out of the code object.
# Measured on MI355X (tools/concurrency_probe.py, profiles/r02_packed_fp32_concurrency.md): a wave executing them
# occasionally gets a wrong result while waves of an MFMA kernel launched on ANOTHER stream share its CU - the LayerNorm
# backward of the text encoder (side stream) differed from run to run underneath the convolution GEMMs until its packed
# ops were gone.  The step time is unchanged (the affected kernels are memory-bound).  packed_fp32_ops() is the check.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-fno-slp-vectorize", "-fno-vectorize"]
FLAGS += os.environ.get("CRIS_EXTRA_HIPCC_FLAGS", "").split()      # experiments (-D switches); part of the build stamp
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + ["common.h", "gemm_common.h", "p2p_ll.h", "jpeg_core.h", os.path.join("..", "..", "..", "include", "cris_hip.h")]:
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _current(dig):
    return os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read() == dig


def build(force=False, verbose=False):
    """Content-stamped (sha256 of sources + flags), so a snapshot with arbitrary mtimes does not rebuild.  Several ranks
    of one node may call this at once (bench.py under torch.distributed.run): an flock serialises them, the first one
    compiles, the others find the stamp current; the library is linked to a temporary name and renamed into place."""
    dig = _digest()
    if not force and _current(dig):
        return LIB
    import fcntl
    with open(os.path.join(HERE, ".build_lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _current(dig):
                return LIB
            return _compile(dig, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def packed_fp32_ops(lib=LIB):
    """{kernel symbol: count} of v_pk_{mul,add,fma}_f32 instructions in the gfx950 code objects of `lib` (None when
    llvm-objdump is not available)"""
    import re
    import shutil
    import tempfile
    if not os.path.exists(OBJDUMP):
        return None
    found = {}
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([OBJDUMP, "--offloading", so], cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(td)):
            if "gfx950" not in f:
                continue
            asm = subprocess.run([OBJDUMP, "-d", os.path.join(td, f)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
            name = "?"
            for line in asm.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
                if m:
                    name = m.group(1)
                elif re.search(r"\bv_pk_(mul|add|fma)_f32\b", line):
                    found[name] = found.get(name, 0) + 1
    return found


def _compile(dig, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(HERE, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(HERE, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
        if verbose and out:
            print(out.decode())
    tmp = LIB + ".tmp.%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stdout.decode())
    bad = packed_fp32_ops(tmp)
    if bad:
        os.remove(tmp)
        raise RuntimeError("packed-FP32 VALU instructions in the code object (see FLAGS): %r" % bad)
    if os.path.exists(STAMP):
        os.remove(STAMP)
    os.replace(tmp, LIB)
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
