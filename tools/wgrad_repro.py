"""Run-to-run identity of the weight-gradient kernels on a few shapes (a schedule hazard shows as rare differing elements)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import ops                 # noqa: E402
from cris.pytorch_amd.ops import Geom            # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
for (B, hw, C, N, k, splits) in ((8, 52, 128, 256, 3, 1), (8, 52, 128, 256, 3, None), (2, 26, 128, 256, 3, 1), (8, 26, 512, 512, 1, 1), (8, 104, 64, 256, 1, 4)):
    g = Geom(B, hw, hw, C, k, k, 1, k // 2)
    torch.manual_seed(0)
    X = torch.randn(g.M, C, device=dev).to(bf)
    dY = torch.randn(g.M, N, device=dev).to(bf)
    outs = []
    for r in range(12):
        dW = torch.full((N, g.K), float("nan"), device=dev)
        db = torch.full((N,), float("nan"), device=dev)
        ops.conv_wgrad(dY, X, g, N, dW, splits=splits, dbias=db)
        torch.cuda.synchronize()
        outs.append((dW, db))
    nd = [int((o[0] != outs[0][0]).sum()) for o in outs[1:]]
    nb = [int((o[1] != outs[0][1]).sum()) for o in outs[1:]]
    if k == 1:
        ref = dY.float().t() @ X.float()
        err = float((outs[0][0] - ref).norm() / ref.norm())
    else:
        err = float("nan")
    bad = (outs[1][0] != outs[0][0]).nonzero()[:6].tolist() if nd[0] else []
    print("REPRO M%d N%d K%d k%d splits=%s: differing elements per rerun %s, bias %s, rel err vs fp32 %.2e %s" % (g.M, N, g.K, k, splits, nd, nb, err, bad), flush=True)
