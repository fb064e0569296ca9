"""Weight-gradient kernels on the step's large shapes, standalone (HIP graph of `reps` launches each).  The tile kernel is
chosen by the library (CRIS_WGRAD8=0/1 is read once per process): run the script once per setting.
    CRIS_WGRAD8=1 python tools/wgrad_bench.py"""
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
SHAPES = [(104, 512, 256, 3), (52, 512, 512, 3), (52, 512, 256, 3), (52, 256, 256, 3), (104, 128, 128, 3), (104, 256, 256, 1),
          (26, 512, 512, 3), (26, 1024, 512, 3), (26, 512, 2048, 1), (13, 2048, 2048, 1)]     # (HW, C, N, k), batch 8
# `--small`: the small-N / small-K weight gradients of the 346112 / 86528 / 21632-pixel maps (stem, layer1, layer2, decoder
# projections; 0.83 ms per step at 35 - 220 TFLOP/s in profiles/r04_gemm_shapes.tsv) - memory-bound shapes: the table also gives
# the operand bytes over the time (dY + X read once, dW written once)
SMALL = [(104, 64, 256, 1), (104, 256, 64, 1), (104, 64, 64, 1), (104, 64, 64, 3), (104, 256, 128, 1), (104, 256, 256, 1), (104, 128, 128, 3),
         (208, 32, 32, 3), (208, 32, 64, 3), (208, 32, 32, 1),
         (52, 128, 512, 1), (52, 512, 128, 1), (52, 128, 128, 3), (52, 256, 512, 1), (52, 512, 256, 1)]
if "--small" in sys.argv:
    SHAPES = SMALL


def timed(hw, C, N, k, reps=10):
    g = Geom(8, hw, hw, C, k, k, 1, k // 2)
    X = torch.randn(g.M, C, device=dev).to(bf)
    dY = torch.randn(g.M, N, device=dev).to(bf)
    dW = torch.empty(N, g.K, device=dev)
    for _ in range(2):
        ops.conv_wgrad(dY, X, g, N, dW)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            ops.conv_wgrad(dY, X, g, N, dW)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return statistics.median(ts), g


for (hw, C, N, k) in SHAPES:
    t, g = timed(hw, C, N, k)
    print("WGRAD8=%s BLOCKS=%s M%d N%d K%d k%d : %.1f us  %.0f TFLOP/s  %.0f GB/s operands (incl. split reduction)" % (
        os.environ.get("CRIS_WGRAD8", "1"), os.environ.get("CRIS_WGRAD_BLOCKS", "512"), g.M, N, g.K, k, t, 2.0 * g.M * N * g.K / t / 1e6,
        (2.0 * g.M * (N + C) + 4.0 * N * g.K) / t / 1e3), flush=True)
