"""What would cross-block split-K on the 8-wave 128x128 tile buy the mid-size GEMMs (round-5 review, item 3)?  Measured with
the kernels that exist, before building the new one:

  t_auto    the problem as the step runs it today (automatic tile), `reps` launches back to back in one HIP graph
  t_main    the SAME number of blocks, K-tiles per block and operand bytes as a split-K launch with S splits would have: the
            8-wave 128x128 tile on a problem of S*M rows and K/S reduction length (a 1x1 problem over S*M pixels - a 3x3
            problem's im2col gather hits L2 the same way: the activation panel is L2-resident at these sizes).  It even
            includes a lean epilogue with statistics - MORE work than a split's raw fp32 slab store
  t_finish  a finishing launch: S fp32 slabs [M, N] summed in order, rounded to bf16 and written - timed as ONE fused streaming
            launch (cris_sum_slabs does not exist: torch's sum(0) + a cast are two launches; their sum is the upper estimate,
            the larger of the two the lower one; both are printed)
  split-K estimate = t_main + t_finish  against  t_auto

    python tools/splitk_emulation.py [--tsv out.tsv]
"""
import argparse
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
HW = {5408: 26, 1352: 13}
# (M, N, K, k, launches per step) - the M <= 5408 / K >= 2048 rows of profiles/r05_gemm_variants.tsv
SHAPES = [(5408, 512, 4608, 3, 8), (5408, 256, 2304, 3, 10), (1352, 2048, 2048, 1, 6), (5408, 512, 2048, 1, 6), (1352, 512, 4608, 3, 4),
          (5408, 1024, 4608, 3, 1), (1352, 512, 2048, 1, 5), (5408, 512, 9216, 3, 1), (1352, 1024, 2048, 1, 3), (1352, 512, 9216, 3, 1),
          (1352, 1024, 4608, 3, 1)]


def graph_time(fn, reps, rounds):
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


def splits_for(M, N, K):
    tiles = -(-M // 128) * -(-N // 128)
    s = 256 // tiles                       # one block per CU (160 KB of LDS per block): never more blocks than CUs
    while s > 1 and (K // 64) // s < 8:    # at least 8 K-tiles per split
        s -= 1
    return max(1, min(s, 8)), tiles


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--tsv", default=None)
    args = ap.parse_args()
    rows = []
    tot_auto = tot_sk = 0.0
    for (M, N, K, k, cnt) in SHAPES:
        S, tiles = splits_for(M, N, K)
        C = K // (k * k)
        hw = HW[M]
        g = Geom(8, hw, hw, C, k, k, 1, k // 2)
        A = torch.randn(M, C, device=dev).to(bf)
        W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
        out = torch.empty(M, N, device=dev, dtype=bf)
        t_auto = graph_time(lambda: ops.conv_gemm(A, W, g, N, out=out, stats=True), args.reps, args.rounds)
        if S < 2:
            print("SPLITK M%d N%d K%d k%d x%d | %d tiles: no room for a split | auto %.1f us" % (M, N, K, k, cnt, tiles, t_auto), flush=True)
            rows.append((M, N, K, k, cnt, tiles, 1, t_auto, None, None, None))
            tot_auto += cnt * t_auto
            tot_sk += cnt * t_auto
            continue
        Ks = (K // 64 + S - 1) // S * 64
        Ms = M * S
        g2 = Geom.linear(Ms, Ks)
        A2 = torch.randn(Ms, Ks, device=dev).to(bf)
        W2 = (torch.randn(N, Ks, device=dev) * 0.05).to(bf)
        out2 = torch.empty(Ms, N, device=dev, dtype=bf)
        t_main = graph_time(lambda: ops.conv_gemm(A2, W2, g2, N, out=out2, stats=True, variant="8w128x128"), args.reps, args.rounds)
        slabs = torch.randn(S, M, N, device=dev)
        acc = torch.empty(M, N, device=dev)
        t_sum = graph_time(lambda: torch.sum(slabs, dim=0, out=acc), args.reps, args.rounds)
        t_cast = graph_time(lambda: out.copy_(acc), args.reps, args.rounds)
        lo, hi = t_main + max(t_sum, t_cast), t_main + t_sum + t_cast
        print("SPLITK M%d N%d K%d k%d x%d | %d tiles x %d splits = %d blocks, %d K-tiles each | auto %.1f us | main %.1f + finish %.1f..%.1f = "
              "%.1f..%.1f us | gain per launch %.1f..%.1f us" % (M, N, K, k, cnt, tiles, S, tiles * S, Ks // 64, t_auto, t_main, max(t_sum, t_cast),
                                                               t_sum + t_cast, lo, hi, t_auto - hi, t_auto - lo), flush=True)
        rows.append((M, N, K, k, cnt, tiles, S, t_auto, t_main, max(t_sum, t_cast), t_sum + t_cast))
        tot_auto += cnt * t_auto
        tot_sk += cnt * min(t_auto, lo)          # the optimistic end: the finish as ONE launch, split only where it wins
        del A, W, out, A2, W2, out2, slabs, acc
        torch.cuda.empty_cache()
    print("SPLITK per step over these %d launches: today %.3f ms, split-K where it wins (optimistic finish) %.3f ms: %.3f ms to gain"
          % (sum(r[4] for r in rows), tot_auto / 1e3, tot_sk / 1e3, (tot_auto - tot_sk) / 1e3))
    if args.tsv:
        with open(args.tsv, "w") as f:
            f.write("M\tN\tK\tk\tlaunches\ttiles128\tsplits\tauto_us\tmain_us\tfinish_lo_us\tfinish_hi_us\n")
            for r in rows:
                f.write("\t".join("-" if x is None else ("%.1f" % x if isinstance(x, float) else str(x)) for x in r) + "\n")


if __name__ == "__main__":
    main()
