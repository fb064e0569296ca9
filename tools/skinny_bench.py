"""Skinny GEMM variants (M = 136 rows: the text encoder's linears) timed standalone: `reps` launches in one HIP graph.
    python tools/skinny_bench.py"""
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


def timed(M, N, K, variant, reps=20):
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    out = torch.empty(M, N, device=dev)
    g = Geom.linear(M, K)

    def launch():
        ops.conv_gemm(A, W, g, N, bias=bias, resid=res, out=out, variant=variant)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            launch()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return statistics.median(ts)


for (M, N, K) in ((136, 512, 2048), (136, 512, 512), (136, 2048, 512), (136, 1536, 512), (136, 512, 1536)):
    row = {v: timed(M, N, K, v) for v in ("skinny9", "skinny9s", "64x64")}
    print("SKINNY M%d N%d K%d | " % (M, N, K) + "  ".join("%s %.1f us (%.1f TF)" % (v, t, 2.0 * M * N * K / t / 1e6) for v, t in row.items()), flush=True)
