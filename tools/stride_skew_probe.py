"""Does the ROW STRIDE of the GEMM operands matter (L2 channel mapping)?  Activations [M][C] and weights [N][K] with power-of-two
row strides (1024 B at C = 512, 9216 B at K = 4608) may land a tile's rows on few L2 channels.  The same problems with the row
stride padded by 64 / 32 elements (128 / 64 B): lda = C + pad, ldb = K + pad.   python tools/stride_skew_probe.py"""
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
HW = {86528: 104, 21632: 52, 5408: 26, 1352: 13}
SHAPES = [(5408, 512, 4608, 3), (5408, 512, 512, 1), (5408, 1024, 256, 1), (5408, 256, 1024, 1), (5408, 256, 2304, 3), (5408, 512, 2048, 1),
          (1352, 512, 4608, 3), (1352, 2048, 512, 1), (21632, 512, 4608, 3), (21632, 128, 512, 1), (21632, 512, 128, 1), (86528, 256, 64, 1),
          (86528, 64, 256, 1), (86528, 256, 4608, 3)]


def graph_time(fn, reps=10, rounds=7):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return statistics.median(ts)


for (M, N, K, k) in SHAPES:
    C = K // (k * k)
    hw = HW[M]
    g = Geom(8, hw, hw, C, k, k, 1, k // 2)
    res = []
    for pa, pb, pc in ((0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (64, 64, 64), (32, 32, 32), (8, 8, 8)):
        A = torch.randn(M, C + pa, device=dev).to(bf)
        W = (torch.randn(N, K + pb, device=dev) * 0.05).to(bf)
        out = torch.empty(M, N + pc, device=dev, dtype=bf)
        t = graph_time(lambda: ops.conv_gemm(A, W, g, N, lda=C + pa, ldb=K + pb, out=out, ldc=N + pc, stats=True))
        res.append(((pa, pb, pc), t))
        del A, W, out
    base = res[0][1]
    print("SKEW M%d N%d K%d k%d | plain %.1f us | " % (M, N, K, k, base) + "  ".join("A+%d W+%d O+%d: %.1f (%+.0f%%)" % (p[0], p[1], p[2], t, 100 * (t / base - 1)) for p, t in res[1:]), flush=True)
    torch.cuda.empty_cache()
