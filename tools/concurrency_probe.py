"""Two-stream probe: the text encoder's backward pattern (skinny GEMM -> LayerNorm backward) repeated from FIXED inputs on a
side stream while the launch stream runs a load of large kernels.  Every repetition must give bit-identical outputs.

This is how the run-to-run differences of the text-encoder gradients were traced (profiles/r02_packed_fp32_concurrency.md):
with the LayerNorm backward compiled with packed-FP32 VALU instructions (v_pk_mul/add/fma_f32 - hipcc's SLP vectoriser
makes them) its row reduction sum(a * xhat) came out wrong in a few waves per few hundred launches whenever an MFMA kernel of
this library (conv GEMM, weight gradient, skinny GEMM) ran on the other stream; never with an elementwise load, never with
the packed ops compiled out (csrc/build.py FLAGS).

    python tools/concurrency_probe.py [none|gemm|wgrad|skinny|bn|fill|torch] [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import ops                 # noqa: E402
from cris.pytorch_amd.ops import Geom            # noqa: E402

LOADS = ("none", "gemm", "wgrad", "skinny", "bn", "fill", "torch")


def run(load="gemm", iters=400, verbose=False):
    """-> (number of GEMM outputs, number of LayerNorm-backward outputs) that differ from the first repetition's"""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    bf = torch.bfloat16
    A = (torch.randn(136, 1536, generator=g) * 1e-3).to(dev, bf)
    W = (torch.randn(512, 1536, generator=g) * 0.05).to(dev, bf)
    x = torch.randn(136, 512, generator=g).to(dev)
    gamma = (1 + 0.1 * torch.randn(512, generator=g)).to(dev)
    beta = torch.zeros(512, device=dev)
    mean, rstd = torch.empty(136, device=dev), torch.empty(136, device=dev)
    y = torch.empty(136, 512, device=dev, dtype=bf)
    ops.ln_fwd(x, gamma, beta, 136, 512, mean, rstd, y=y)
    Xb = (torch.randn(8 * 104 * 104, 64, generator=g)).to(dev, bf)
    Wb = (torch.randn(64, 9 * 64, generator=g) * 0.05).to(dev, bf)
    Yb = torch.empty(8 * 104 * 104, 64, device=dev, dtype=bf)
    Zb = torch.empty(8 * 104 * 104, 64, device=dev, dtype=bf)
    sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    dWb = torch.empty(64, 9 * 64, device=dev)
    As = (torch.randn(136, 2048, generator=g) * 1e-2).to(dev, bf)
    Ws = (torch.randn(16384, 2048, generator=g) * 0.05).to(dev, bf)
    Ys = torch.empty(136, 16384, device=dev, dtype=bf)
    Tm = torch.randn(4096, 4096, device=dev, dtype=bf)
    big = torch.empty(64 << 20, device=dev)
    torch.cuda.synchronize()
    side, main = torch.cuda.Stream(), torch.cuda.current_stream()
    hs, dxs = [], []
    torch.cuda._sleep(40000000)                  # hold the device back: both streams' work queues up, then runs concurrently
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for _ in range(iters):
            h = torch.empty(136, 512, device=dev, dtype=bf)
            ops.conv_gemm(A, W, Geom.linear(136, 1536), 512, out=h)
            dx = torch.empty(136, 512, device=dev)
            ops.ln_bwd(x, gamma, mean, rstd, 136, 512, dx, dy=h)
            hs.append(h)
            dxs.append(dx)
    for _ in range(iters // 4):
        if load == "gemm":            # LDS-DMA ring + MFMA
            ops.conv_gemm(Xb, Wb, Geom(8, 104, 104, 64, 3, 3, 1, 1), 64, out=Yb)
        elif load == "wgrad":         # LDS-DMA + transposing LDS reads + MFMA
            ops.conv_wgrad(Yb, Xb, Geom(8, 104, 104, 64, 3, 3, 1, 1), 64, dWb)
        elif load == "skinny":        # MFMA from registers, no LDS-DMA
            ops.conv_gemm(As, Ws, Geom.linear(136, 2048), 16384, out=Ys)
        elif load == "bn":            # streaming elementwise kernel: no LDS, no MFMA
            ops.bn_apply(Xb, sc, sh, Zb, 8, 104, 104, 64)
        elif load == "fill":          # torch's own streaming kernel
            big.fill_(1.0)
        elif load == "torch":         # hipBLASLt GEMM
            Tm @ Tm
    main.wait_stream(side)
    torch.cuda.synchronize()
    bad_h = [i for i in range(iters) if not torch.equal(hs[i], hs[0])]
    bad_dx = [i for i in range(iters) if not torch.equal(dxs[i], dxs[0])]
    if verbose:
        print("load", load, "iters", iters, "| GEMM outputs differing from the first:", len(bad_h), bad_h[:8],
              "| ln_bwd outputs differing:", len(bad_dx), bad_dx[:8])
        xh = (x - mean[:, None]) * rstd[:, None]
        a = hs[0].float() * gamma[None, :]
        o = rstd[:, None] * (a - a.mean(1, keepdim=True) - xh * (a * xh).mean(1, keepdim=True))
        for i in bad_dx[:3]:
            d = (dxs[i] - dxs[0]).abs()
            rows = torch.nonzero(d.sum(1) > 0).flatten().tolist()
            print("   iteration", i, "rows", rows, "max|d| %.3e" % float(d.max()), "h equal:", torch.equal(hs[i], hs[0]))
            for r in rows[:2]:      # which part of dx = rstd * (a - mean(a) - xhat * mean(a * xhat)) explains the difference?
                dd = (dxs[i][r] - dxs[0][r]).double()
                basis = torch.stack([o[r], xh[r], torch.ones_like(xh[r])], 1).double()
                sol = torch.linalg.lstsq(basis, dd[:, None]).solution.flatten()
                res = dd - basis @ sol
                print("      row %d |diff| %.3e = %s . (dx, xhat, 1) + residual %.3e;  mean(a) %.3e  mean(a*xhat) %.3e" % (
                    r, float(dd.norm()), ["%.2e" % float(c) for c in sol], float(res.norm()), float(a[r].mean()),
                    float((a[r] * xh[r]).mean())))
    return len(bad_h), len(bad_dx)


if __name__ == "__main__":
    run(sys.argv[1] if len(sys.argv) > 1 else "gemm", int(sys.argv[2]) if len(sys.argv) > 2 else 400, verbose=True)
