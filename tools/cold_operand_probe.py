"""Which operand's coldness costs a mid-size GEMM its in-step time?  (In the step the 64x64-tile launches take ~17.6 us where a
back-to-back loop on one set of operands measures ~9.8.)  One HIP graph of `reps` launches of one shape, cycling through `nw`
weight copies and `na` activation copies: 1 copy = warm in L2; 64 copies of a 0.5 MB weight = out of the 4 MB L2 of an XCD but
inside the 256 MB Infinity Cache; 1024 copies = out of both.  Activations likewise (5.5 MB each).
    python tools/cold_operand_probe.py"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import ops                 # noqa: E402
from cris.pytorch_amd.ops import Geom            # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
HW = {5408: 26, 1352: 13, 21632: 52}


def run(M, N, K, k, nw, na, reps=128, rounds=5):
    C = K // (k * k)
    hw = HW[M]
    g = Geom(8, hw, hw, C, k, k, 1, k // 2)
    As = [torch.randn(M, C, device=dev).to(bf) for _ in range(na)]
    Ws = [(torch.randn(N, K, device=dev) * 0.05).to(bf) for _ in range(nw)]
    outs = [torch.empty(M, N, device=dev, dtype=bf) for _ in range(min(na, 8))]
    def body():
        for i in range(reps):
            ops.conv_gemm(As[i % na], Ws[(i * 7) % nw], g, N, out=outs[i % len(outs)], stats=True)
    body()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        body()
    gr.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    del As, Ws, outs, gr
    torch.cuda.empty_cache()
    return statistics.median(ts)


for (M, N, K, k) in ((5408, 512, 512, 1), (5408, 1024, 256, 1), (5408, 256, 1024, 1), (5408, 256, 2304, 3), (1352, 2048, 512, 1), (5408, 512, 4608, 3)):
    wbytes = N * K * 2 / 1e6
    abytes = M * (K // (k * k)) * 2 / 1e6
    n_l2 = max(2, int(64 / wbytes) + 1)            # > 32 MB of weights in rotation: out of every L2
    n_mall = max(2, int(600 / wbytes) + 1)         # > 512 MB: out of the Infinity Cache too
    a_mall = max(2, int(600 / abytes) + 1)
    t_warm = run(M, N, K, k, 1, 1)
    t_w_l2 = run(M, N, K, k, n_l2, 1)
    t_w_hbm = run(M, N, K, k, min(n_mall, 1200), 1)
    t_a_hbm = run(M, N, K, k, 1, min(a_mall, 200))
    t_both = run(M, N, K, k, min(n_mall, 1200), min(a_mall, 200))
    print("COLD M%d N%d K%d k%d (W %.1f MB, A %.1f MB) | warm %.1f us | W out of L2 %.1f | W from HBM %.1f | A from HBM %.1f | both from HBM %.1f"
          % (M, N, K, k, wbytes, abytes, t_warm, t_w_l2, t_w_hbm, t_a_hbm, t_both), flush=True)
