"""Standalone timing of the LayerNorm kernels at the shapes the CRIS-R50 step runs them (decoder: 5408 rows; text: 136 rows).
    python tools/ln_bench.py            (on a GPU box)
Prints us per launch and the effective memory rate for each site."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cris.pytorch_amd import ops  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def timeit(fn, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def site(name, rows, C, x_f32, dy, dypos, dout, dx_f32, accum, in_relu=False, in_drop=False, out_drop=False):
    x = torch.randn(rows, C, device=DEV).to(torch.float32 if x_f32 else BF)
    gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    y = torch.empty(rows, C, dtype=BF, device=DEV)
    ind = ops.Drop(0.1, 1, 1) if in_drop else ops.NO_DROP
    outd = ops.Drop(0.1, 1, 2) if out_drop else ops.NO_DROP
    resid = torch.randn(rows, C, device=DEV) if dout else None
    outs = torch.empty(rows, C, device=DEV) if dout else None
    fwd = lambda: ops.ln_fwd(x, gamma, beta, rows, C, mean, rstd, y=y if dy else None, resid=resid, out_f32=outs,  # noqa: E731
                             in_relu=in_relu, in_drop=ind, out_drop=outd)
    t_f = timeit(fwd)
    g_y = torch.randn(rows, C, device=DEV).to(BF) if dy else None
    g_p = torch.randn(rows, C, device=DEV).to(BF) if dypos else None
    g_o = torch.randn(rows, C, device=DEV) if dout else None
    dx = torch.zeros(rows, C, device=DEV, dtype=torch.float32 if dx_f32 else BF)
    dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    q = ops.SumQueue()
    bwd = lambda: (ops.ln_bwd(x, gamma, mean, rstd, rows, C, dx, dy=g_y, dypos=g_p, dout_f32=g_o, dgamma=dg, dbeta=db,  # noqa: E731
                              dx_accum=accum, in_relu=in_relu, in_drop=ind, out_drop=outd, queue=q), q.items.clear())
    t_b = timeit(bwd)
    e = rows * C
    bytes_b = e * ((4 if x_f32 else 2) + (2 if dy else 0) + (2 if dypos else 0) + (4 if dout else 0) + (4 if dx_f32 else 2) * (2 if accum else 1))
    print("LN %-18s rows %5d C %4d : fwd %6.1f us   bwd %6.1f us (%.2f TB/s of %d MB)" % (name, rows, C, t_f, t_b, bytes_b / t_b / 1e6,
                                                                                        bytes_b >> 20))


if __name__ == "__main__":
    site("norm1", 5408, 512, True, True, True, False, True, True)
    site("self_attn_norm", 5408, 512, False, False, False, True, False, False, out_drop=True)
    site("norm3", 5408, 512, True, True, False, False, True, True)
    site("ffn.3", 5408, 2048, False, True, False, False, False, False, in_relu=True, in_drop=True)
    site("text ln", 136, 512, True, True, False, False, True, True)
