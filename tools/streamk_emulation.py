"""What would stream-K (the K-tiles of T < 256 output tiles dealt out evenly to 256 one-block-per-CU blocks, partial tiles fixed up
through fp32 slabs) buy the 8-wave launches that leave a third of the chip idle (profiles/r06/gemm8_round_quantisation.md)?
Measured with the kernel that exists: the same tile variant on a problem with exactly 256 tiles and K' = K * T / 256 (the K-tiles one
block would get), i.e. the same blocks x K-tiles x operand bytes per block with EVERY CU busy - plus nothing for the fix-up (a
shared tile's partial is 64 - 256 KB written and read once through memory + one flag round trip: +3 ... 6 us per launch, not
measured here).  t_now = the problem as it runs today.

    python tools/streamk_emulation.py"""
import math
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
HW = {21632: 52, 5408: 26, 1352: 13}
TILE = {"8w256x256": (256, 256), "8w128x256": (128, 256), "8w128x128": (128, 128)}
# (M, N, K, k, launches per step, variant) - the rows of gemm8_round_quantisation.md with a busy fraction of ~0.67
SHAPES = [(5408, 512, 512, 1, 15, "8w128x128"), (21632, 512, 4608, 3, 2, "8w256x256"), (5408, 2048, 512, 1, 6, "8w256x256"),
          (5408, 512, 2048, 1, 6, "8w128x128"), (5408, 1024, 256, 1, 11, "8w128x256"), (5408, 512, 4608, 3, 4, "8w128x128"),
          (21632, 128, 1152, 3, 6, "8w128x128"), (21632, 128, 512, 1, 7, "8w128x128"), (21632, 256, 2304, 3, 2, "8w128x256"),
          (1352, 2048, 2048, 1, 3, "8w128x128"), (5408, 512, 9216, 3, 1, "8w128x128"), (21632, 512, 2304, 3, 1, "8w256x256"),
          (21632, 256, 4608, 3, 1, "8w128x256")]


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


tot_now = tot_sk = 0.0
for (M, N, K, k, cnt, v) in SHAPES:
    bm, bn = TILE[v]
    T = math.ceil(M / bm) * math.ceil(N / bn)
    C = K // (k * k)
    hw = HW[M]
    g = Geom(8, hw, hw, C, k, k, 1, k // 2)
    A = torch.randn(M, C, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    out = torch.empty(M, N, device=dev, dtype=bf)
    t_now = graph_time(lambda: ops.conv_gemm(A, W, g, N, out=out, stats=True, variant=v))
    # 256 tiles: keep the column tiles, 256 / (N / bn) row tiles; K' = the K-tiles one of 256 blocks would get
    ct = math.ceil(N / bn)
    M2 = (256 // ct) * bm
    K2 = max(64, int(math.ceil(K / 64 * T / 256.0)) * 64)
    A2 = torch.randn(M2, K2, device=dev).to(bf)
    W2 = (torch.randn(N, K2, device=dev) * 0.05).to(bf)
    out2 = torch.empty(M2, N, device=dev, dtype=bf)
    t_sk = graph_time(lambda: ops.conv_gemm(A2, W2, Geom.linear(M2, K2), N, out=out2, stats=True, variant=v))
    print("STREAMK M%d N%d K%d k%d x%d %s | %d tiles (%.2f of the CUs) | now %.1f us | 256 blocks x %d K-tiles: %.1f us | gain before fix-up %.1f us"
          % (M, N, K, k, cnt, v, T, T / 256.0, t_now, K2 // 64, t_sk, t_now - t_sk), flush=True)
    tot_now += cnt * t_now
    tot_sk += cnt * min(t_now, t_sk + 4.0)             # + ~4 us of fix-up where a split is worth it at all
    del A, W, out, A2, W2, out2
    torch.cuda.empty_cache()
print("STREAMK per step over these launches: now %.3f ms; stream-K incl. ~4 us of fix-up per launch, only where it wins: %.3f ms; to gain %.3f ms"
      % (tot_now / 1e3, tot_sk / 1e3, (tot_now - tot_sk) / 1e3))
