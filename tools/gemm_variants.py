"""Per-shape A/B of the conv_gemm tile variants (cris_conv_gemm_variant): every GEMM shape of the benchmarked training step
(profiles/r02_gemm_shapes.tsv: forward + input-gradient launches) timed with each applicable tile variant - lean epilogue
with BatchNorm statistics, random bf16 operands, `reps` back-to-back launches inside one HIP graph, variants interleaved
round-robin over `rounds` rounds (median reported).  Prints one line per shape and a summary of what the current automatic
choice loses against the best variant.

    python tools/gemm_variants.py [--min-m 5000] [--rounds 5] [--tsv out.tsv]
"""
import argparse
import os
import re
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import ops                 # noqa: E402
from cris.pytorch_amd.ops import Geom            # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
HW = {346112: 208, 86528: 104, 21632: 52, 5408: 26, 1352: 13}


def shapes_of_step():
    """[(M, N, K, k, launches per step)] from the committed shape table"""
    out = []
    for line in open(os.path.join(ROOT, "profiles", "r02_gemm_shapes.tsv")):
        m = re.match(r"conv_gemm\tM(\d+) N(\d+) K(\d+) k(\d)\t([\d.]+)", line)
        if m:
            out.append((int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5))))
    return out


def make_runner(M, N, K, k, variant, stats, reps):
    C = K // (k * k)
    hw = HW[M]
    g = Geom(8, hw, hw, C, k, k, 1, k // 2)
    A = torch.randn(M, C, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    out = torch.empty(M, N, device=dev, dtype=bf)
    for _ in range(2):
        ops.conv_gemm(A, W, g, N, out=out, stats=stats, variant=variant)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            ops.conv_gemm(A, W, g, N, out=out, stats=stats, variant=variant)
    gr.replay()
    torch.cuda.synchronize()

    def run():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    return run, (A, W, out, gr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-m", type=int, default=1000)
    ap.add_argument("--min-k", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--variants", default="")
    ap.add_argument("--tsv", default=None)
    args = ap.parse_args()
    names = ops.gemm_variants()
    want = [v for v in (args.variants.split(",") if args.variants else names) if not v.startswith("skinny")]
    rows = []
    tot_auto = tot_best = 0.0
    for (M, N, K, k, cnt) in shapes_of_step():
        if M < args.min_m or K < args.min_k or M not in HW:
            continue
        C = K // (k * k)
        runners = {}
        for v in ["auto"] + want:
            if v != "auto" and v.startswith("8w") and C % 64:
                continue
            if v != "auto" and v.startswith("8w") and v != "8w128x128" and (M < 5000 or (v != "8w256x128" and N <= 128) or (v == "8w256x128" and N > 256)):
                continue
            if v == "8w128x128" and (M > 30000 or N <= 64):
                continue
            try:
                runners[v] = make_runner(M, N, K, k, -1 if v == "auto" else v, True, args.reps)
            except Exception as e:      # noqa: BLE001
                print("skip", v, M, N, K, repr(e)[:100])
        ts = {v: [] for v in runners}
        for _ in range(args.rounds):
            for v, (run, _) in runners.items():
                ts[v].append(run())
        med = {v: statistics.median(t) for v, t in ts.items()}
        best = min((v for v in med if v != "auto"), key=lambda v: med[v])
        fl = 2.0 * M * N * K
        tot_auto += med["auto"] * cnt
        tot_best += med[best] * cnt
        line = "M%d N%d K%d k%d x%.0f | auto %.1fus %.0fTF | best %s %.1fus %.0fTF | " % (
            M, N, K, k, cnt, med["auto"], fl / med["auto"] / 1e6, best, med[best], fl / med[best] / 1e6)
        line += " ".join("%s=%.1f" % (v, med[v]) for v in med if v != "auto")
        print("GEMMVAR", line, flush=True)
        rows.append((M, N, K, k, cnt, med))
        del runners
        torch.cuda.empty_cache()
    print("GEMMVAR total per step: auto %.3f ms, best-per-shape %.3f ms" % (tot_auto / 1e3, tot_best / 1e3))
    if args.tsv:
        with open(args.tsv, "w") as f:
            f.write("M\tN\tK\tk\tlaunches\t" + "\t".join(["auto"] + want) + "\n")
            for (M, N, K, k, cnt, med) in rows:
                f.write("%d\t%d\t%d\t%d\t%.0f\t" % (M, N, K, k, cnt) + "\t".join("%.1f" % med[v] if v in med else "-" for v in ["auto"] + want) + "\n")


if __name__ == "__main__":
    main()
