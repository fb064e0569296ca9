"""How much of the 8-wave GEMM launches' time is round quantisation (tiles against the 256 CUs)?  For every conv GEMM shape of a
bench shape table (profiles/rNN_gemm_shapes.tsv) the tile variant the library picks (cris_conv_gemm_plan: host code, no GPU needed),
its tile count, the rounds of 256 one-block-per-CU tiles, and what the launch would take if every CU were busy all the time.

    python tools/gemm_round_quantisation.py profiles/r06_gemm_shapes.tsv > profiles/r06/gemm8_round_quantisation.md"""
import ctypes as C
import math
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import hip      # noqa: E402

lib = hip.load()
names = [lib.cris_conv_gemm_variant_name(i).decode() for i in range(lib.cris_conv_gemm_num_variants())]
HW = {346112: 208, 86528: 104, 21632: 52, 5408: 26, 1352: 13}
TILE = {"8w256x256": (256, 256), "8w256x128": (256, 128), "8w128x256": (128, 256), "8w128x128": (128, 128)}
rows = []
for line in open(sys.argv[1]):
    m = re.match(r"conv_gemm\tM(\d+) N(\d+) K(\d+) k(\d)\t([\d.]+)\t([\d.]+)\t([\d.]+)\t([\d.]+)", line)
    if not m:
        continue
    M, N, K, k = (int(m.group(i)) for i in range(1, 5))
    n, us, tf = float(m.group(5)), float(m.group(7)), float(m.group(8))
    if M not in HW:
        continue
    p = hip.ConvGemmParams()
    hw, Cc = HW[M], K // (k * k)
    p.Bn, p.H, p.W, p.C = 8, hw, hw, Cc
    p.OH, p.OW, p.KH, p.KW, p.stride, p.pad = hw, hw, k, k, 1, k // 2
    p.M, p.N, p.K, p.lda, p.ldb, p.ldc = M, N, K, Cc, K, N
    p.A = p.Wt = p.out = 16                       # (non-null: the plan looks at pointers only to tell the epilogue kind)
    epi = C.c_int(0)
    v = names[lib.cris_conv_gemm_plan(C.byref(p), -1, C.byref(epi))]
    if v in TILE:
        rows.append((M, N, K, k, n, us, tf, v))
print("# Round quantisation of the 8-wave tile launches (one block per CU: 128 - 160 KB of LDS), from `%s`\n" % os.path.basename(sys.argv[1]))
print("| shape | launches / step | variant | tiles | rounds of 256 CUs | busy fraction | us / launch | TFLOP/s | TFLOP/s of the busy CUs |")
print("|---|---|---|---|---|---|---|---|---|")
tot = ideal = 0.0
for M, N, K, k, n, us, tf, v in sorted(rows, key=lambda r: -r[4] * r[5]):
    bm, bn = TILE[v]
    tiles = math.ceil(M / bm) * math.ceil(N / bn)
    rounds = tiles / 256.0
    eff = rounds / math.ceil(rounds)
    tot += n * us
    ideal += n * us * eff
    print("| M%d N%d K%d k%d | %.0f | %s | %d | %.2f | %.2f | %.1f | %.0f | %.0f |" % (M, N, K, k, n, v, tiles, rounds, eff, us, tf, tf / eff))
print("\n%d shapes, %.3f ms per step in these launches (HIP events, eager pass); scaled by the busy fraction: %.3f ms, i.e. %.3f ms per "
      "step of idle CU time.  That is NOT recoverable time: tools/streamk_emulation.py (profiles/r06/streamk_emulation.log) runs the same "
      "tile kernels with all 256 CUs busy on 2 / 3 of the K-tiles each - what stream-K would schedule - and the launches take as long as "
      "before (M 5408 / N 512 / K 4608: 36.6 us on 172 CUs, 39.1 us on 256): these problems are bound by the chip's shared L2 -> LDS "
      "operand path, which 172 blocks already saturate, not by the number of CUs at work." % (len(rows), tot / 1e3, ideal / 1e3, (tot - ideal) / 1e3))
