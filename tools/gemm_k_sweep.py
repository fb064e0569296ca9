"""Where does a mid-size conv GEMM's time go?  Duration against K at fixed M x N, 60 dependent launches inside a HIP graph:
t(K) = a + b*K separates the per-launch fixed cost (dispatch, ring prologue, epilogue with BatchNorm statistics) from the
steady-state rate of the main loop.   python tools/gemm_k_sweep.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import ops                 # noqa: E402
from cris.pytorch_amd.ops import Geom            # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16


def time_gemm(M, N, K, stats, reps=60):
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    out = torch.empty(M, N, device=dev, dtype=bf)
    g = Geom.linear(M, K)
    for _ in range(5):
        ops.conv_gemm(A, W, g, N, out=out, stats=stats)
    torch.cuda.synchronize()
    # the launches are captured into a HIP graph and replayed: eager launches from Python cost ~11-14 us each, more than
    # the kernels measured here
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            ops.conv_gemm(A, W, g, N, out=out, stats=stats)
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps          # us per launch inside a graph (dependent launches, back to back)


for (M, N) in ((5408, 512), (5408, 256), (1352, 1024), (21632, 256), (86528, 256)):
    rows = []
    for K in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
        if M * K > 86528 * 2048:
            continue
        t = time_gemm(M, N, K, True)
        rows.append((K, t))
    # least squares t = a + b K over the K >= 512 points
    pts = [(k, t) for k, t in rows if k >= 512]
    n = len(pts)
    sk, st = sum(k for k, _ in pts), sum(t for _, t in pts)
    skk, skt = sum(k * k for k, _ in pts), sum(k * t for k, t in pts)
    b = (n * skt - sk * st) / (n * skk - sk * sk)
    a = (st - b * sk) / n
    print("M %6d N %5d | " % (M, N) + "  ".join("K%d %.1fus" % (k, t) for k, t in rows))
    print("              fit (K>=512): fixed %.1f us + %.4f us per K element = %.0f TFLOP/s steady state; K=512 launch is %.0f %% fixed cost" % (
        a, b, 2.0 * M * N / b / 1e6, 100.0 * a / (a + b * 512)))
t0 = time_gemm(64, 64, 64, False, reps=200)
print("smallest launch (one 64x64 tile, K 64): %.1f us per graph node" % t0)
