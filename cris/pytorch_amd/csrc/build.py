"""Build libcris_hip.so for gfx950 (hipcc cross-compiles without a GPU).  In-tree output:
cris/pytorch_amd/csrc/libcris_hip.so (git-ignored, shipped to the GPU box by gpurun)."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.hip", "gemm.hip", "wgrad.hip", "norm.hip", "attention.hip", "elementwise.hip", "evalpost.hip", "p2p.hip"]
LIB = os.path.join(HERE, "libcris_hip.so")
STAMP = os.path.join(HERE, ".build_stamp")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + ["common.h", os.path.join("..", "..", "..", "include", "cris_hip.h")]:
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
    if os.path.exists(STAMP):
        os.remove(STAMP)
    os.replace(tmp, LIB)
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
