"""GPU parity tests, kernel by kernel: every C-ABI launcher vs a plain torch fp32 statement of the same
op on the same (bf16-rounded) inputs.  Tolerances: bf16 outputs -> relative L2 error <= 1e-2 (one bf16
rounding of an fp32-accumulated result is 2^-9 ~ 2e-3 per element); fp32 outputs of bf16 MFMA products ->
5e-3; pure fp32 kernels -> 1e-5.  Index/mask decisions (dropout keep, nearest resize, EOT argmax) are
bit exact."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cris.pytorch_amd import hip, ops  # noqa: E402
from cris.pytorch_amd.ops import Geom, Drop  # noqa: E402
from oracle import dropout_hash  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale)


def bf(x):
    return x.to(DEV).to(BF).contiguous()


def relerr(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def check(a, b, tol, what=""):
    e = relerr(a, b)
    assert math.isfinite(e) and e <= tol, "%s rel L2 err %.3e > %.1e" % (what, e, tol)


def keep_mask(seed, stream, shape, p):
    n = int(np.prod(shape))
    return torch.from_numpy(dropout_hash.keep_mask(seed, stream, n, p)).view(*shape).to(DEV)


# ----------------------------------------------------------------------------------------------------
# implicit GEMM
# ----------------------------------------------------------------------------------------------------
def conv_ref(x_nhwc, w, stride, pad):
    """x [B,H,W,C] fp32, w [N,C,KH,KW] fp32 -> [B*OH*OW, N] fp32"""
    y = F.conv2d(x_nhwc.permute(0, 3, 1, 2), w, stride=stride, padding=pad)
    return y.permute(0, 2, 3, 1).reshape(-1, w.shape[0])


def pack_F(w, Cpad=None):
    """[N,C,KH,KW] -> [N, taps*Cpad] (k = tap*Cpad + c)"""
    N, C_, KH, KW = w.shape
    Cpad = Cpad or C_
    out = torch.zeros(N, KH * KW, Cpad)
    out[:, :, :C_] = w.permute(0, 2, 3, 1).reshape(N, KH * KW, C_)
    return out.reshape(N, KH * KW * Cpad)


@pytest.mark.parametrize("case", [
    dict(B=2, H=12, W=12, C=64, N=128, k=1),
    dict(B=2, H=12, W=12, C=64, N=64, k=3),
    dict(B=1, H=20, W=20, C=32, N=32, k=3),          # BN=64 path, K=288 (tail in the K loop)
    dict(B=2, H=9, W=9, C=136, N=72, k=3),           # C not a multiple of 64: chunks straddle taps; N tail
    dict(B=3, H=8, W=10, C=24, N=200, k=3, stride=2),
    dict(B=8, H=1, W=1, C=1024, N=2305, k=1),        # proj.txt shape: M=8, N tail
    dict(B=700, H=1, W=1, C=512, N=1536, k=1),       # linear
    # the shapes of the benchmarked configuration that select the 128x128 tile (>= 448 tiles): proj.vis.3 and proj.vis.1
    dict(B=8, H=104, W=104, C=512, N=256, k=3),      # M=86528 N=256 K=4608: 1352 tiles, 2-deep ring
    dict(B=8, H=52, W=52, C=512, N=512, k=3),        # M=21632 N=512 K=4608: 676 tiles
    dict(B=8, H=104, W=104, C=64, N=256, k=1),       # M=86528 N=256 K=64: one K-step (prologue == whole loop), HBM-bound
])
def test_conv_gemm_plain(case):
    B, H, W, C_, N, k = case["B"], case["H"], case["W"], case["C"], case["N"], case["k"]
    stride = case.get("stride", 1)
    pad = k // 2
    x = rnd(B, H, W, C_).to(BF).float()
    w = (rnd(N, C_, k, k, seed=1) / math.sqrt(C_ * k * k)).to(BF).float()
    g = Geom(B, H, W, C_, k, k, stride, pad)
    out = torch.empty(g.M, N, dtype=BF, device=DEV)
    ops.conv_gemm(bf(x), bf(pack_F(w)), g, N, out=out)
    ref = conv_ref(x, w, stride, pad)
    check(out, ref, 6e-3, "conv_gemm %s" % case)


def test_conv_gemm_epilogues():
    M, K, N = 300, 256, 192
    x = rnd(M, K).to(BF).float()
    w = (rnd(N, K, seed=1) / math.sqrt(K)).to(BF).float()
    bias = rnd(N, seed=2)
    res32 = rnd(M, N, seed=3)
    resbf = rnd(M, N, seed=4).to(BF).float()
    g = Geom.linear(M, K)
    base = x @ w.t() + bias
    # fp32 out, fp32 residual, QuickGELU
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.conv_gemm(bf(x), bf(w), g, N, bias=bias.to(DEV), act=2, resid=res32.to(DEV), out=out)
    check(out, base * torch.sigmoid(1.702 * base) + res32, 3e-3, "quickgelu+resid f32")
    # bf16 out into a column slice of a wider buffer, bf16 residual, relu
    wide = torch.zeros(M, N + 64, dtype=BF, device=DEV)
    ops.conv_gemm(bf(x), bf(w), g, N, bias=bias.to(DEV), act=1, resid=bf(resbf), out=wide, ldc=N + 64, c_coff=64)
    check(wide[:, 64:], torch.relu(base) + resbf, 6e-3, "relu+resid bf16 slice")
    assert float(wide[:, :64].float().abs().max()) == 0.0
    # BatchNorm statistics partials: per 64-row block (sum, M2 about the block mean), merged by bn_finalize
    out2 = torch.empty(M, N, dtype=BF, device=DEV)
    st = ops.conv_gemm(bf(x), bf(w + 0.05), g, N, out=out2, stats=True)
    y = x @ (w + 0.05).to(BF).float().t()
    rows = st.rows_per_part
    for part in range((M + rows - 1) // rows):
        blk = y[part * rows:(part + 1) * rows]
        check(st[0][part], blk.sum(0), 3e-3, "part sum")
        check(st[1][part], ((blk - blk.mean(0)) ** 2).sum(0), 5e-3, "part M2")
    outs = [torch.empty(N, device=DEV) for _ in range(4)]
    ops.bn_finalize(st, M, M, torch.ones(N, device=DEV), torch.zeros(N, device=DEV), None, None, 0.1, 1e-5, N, *outs)
    check(outs[2], y.mean(0), 3e-3, "mean from partials")
    check(outs[3], torch.rsqrt(y.var(0, unbiased=False) + 1e-5), 3e-3, "invstd from partials")
    # dropout (mask is an index op: exact), then residual
    p, seed, stream = 0.1, 1234, 7
    out3 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.conv_gemm(bf(x), bf(w), g, N, bias=bias.to(DEV), resid=res32.to(DEV), out=out3, drop=Drop(p, seed, stream))
    km = keep_mask(seed, stream, (M, N), p).cpu()
    ref = base * km / (1 - p) + res32
    check(out3, ref, 3e-3, "dropout epilogue")
    dropped = (out3.cpu() - res32).abs() < 1e-12
    assert bool((dropped == ~km).all()), "dropout keep decisions differ from the hash oracle"


@pytest.mark.parametrize("case", [
    dict(B=2, H=13, W=13, C=64, N=128, k=1),           # 64x64 tile
    dict(B=2, H=26, W=26, C=256, N=64, k=3),           # N <= 64: 128x64 tile
    dict(B=8, H=26, W=26, C=512, N=512, k=3),          # K >= 4096, M >= 4096: 64x128 tile
    dict(B=8, H=52, W=52, C=128, N=512, k=1),          # M 21632 x N 512: 676 tiles of 128x128
    dict(B=3, H=9, W=9, C=24, N=200, k=3),             # ragged C / N tails
])
def test_conv_gemm_folded_batchnorm_epilogue(case):
    """the inference epilogue (EPI 2 of csrc/gemm.hip; cris/pytorch_amd/infer.py): weights packed with a per-row scale
    (cris_pack_weights row_scale), per-column bias, ReLU before (act 1) / after (act 3) a bf16 residual, no activation with a
    bias (act 0) - against conv2d + eval-mode batch_norm + relu in fp32"""
    B, H, W, C_, N, k = case["B"], case["H"], case["W"], case["C"], case["N"], case["k"]
    pad = k // 2
    x = rnd(B, H, W, C_).to(BF).float()
    w = rnd(N, C_, k, k, seed=1) / math.sqrt(C_ * k * k)
    gamma, beta = 1.0 + 0.3 * rnd(N, seed=2), 0.2 * rnd(N, seed=3)
    rmean, rvar = 0.1 * rnd(N, seed=4), 0.5 + rnd(N, seed=5).abs()
    g = Geom(B, H, W, C_, k, k, 1, pad)
    scale = torch.empty(N, device=DEV)
    shift = torch.empty(N, device=DEV)
    ops.bn_eval_coeffs(gamma.to(DEV), beta.to(DEV), rmean.to(DEV), rvar.to(DEV), 1e-5, N, scale, shift)
    tab = ops.PackTable()
    wdev = w.to(DEV).contiguous()
    wf, _ = tab.add(wdev, N, C_, k * k, Cpad=ops.pad8(C_), want_D=False, row_scale=scale)
    tab.run()
    s_ref = gamma / torch.sqrt(rvar + 1e-5)
    check(scale, s_ref, 1e-6, "eval scale")
    # (the pack's row stride may exceed its k * k * Cpad values: ops.pack_row_stride)
    check(wf[:, :k * k * ops.pad8(C_)].reshape(N, k * k, ops.pad8(C_))[:, :, :C_], (w * s_ref.view(N, 1, 1, 1)).permute(0, 2, 3, 1).reshape(N, k * k, C_), 4e-3, "row-scaled pack")
    xin = x if ops.pad8(C_) == C_ else F.pad(x, (0, ops.pad8(C_) - C_))
    gk = Geom(B, H, W, ops.pad8(C_), k, k, 1, pad)
    y = F.batch_norm(conv_ref(x, w, 1, pad).t().reshape(1, N, -1), rmean, rvar, gamma, beta, False, 0.0, 1e-5).reshape(N, -1).t()
    resid = rnd(g.M, N, seed=6).to(BF).float()
    for act, res, ref in ((1, None, torch.relu(y)), (3, resid, torch.relu(y + resid)), (0, None, y), (1, resid, torch.relu(y) + resid)):
        out = torch.empty(g.M, N, dtype=BF, device=DEV)
        ops.conv_gemm(bf(xin), wf, gk, N, bias=shift, act=act, resid=None if res is None else bf(res), out=out)
        check(out, ref, 8e-3, "folded BN act %d resid %s %s" % (act, res is not None, case))


def test_conv_gemm_general_epilogue_large_tiles():
    """the non-LEAN instantiation of the 128x128 tile (bias + fp32 residual + fp32 out; BN statistics partials of 64 rows)"""
    M, K, N = 21632 * 3, 512, 512                     # 507 x 4 tiles of 128x128
    x = rnd(M, K).to(BF).float()
    w = (rnd(N, K, seed=1) / math.sqrt(K)).to(BF).float()
    bias, res = rnd(N, seed=2), rnd(M, N, seed=3)
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    st = ops.conv_gemm(bf(x), bf(w), Geom.linear(M, K), N, bias=bias.to(DEV), resid=res.to(DEV), out=out, stats=True)
    ref = x @ w.t() + bias + res
    check(out, ref, 3e-3, "128x128 general epilogue")
    assert st.rows_per_part == 64
    outs = [torch.empty(N, device=DEV) for _ in range(4)]
    ops.bn_finalize(st, M, M, torch.ones(N, device=DEV), torch.zeros(N, device=DEV), None, None, 0.1, 1e-5, N, *outs)
    check(outs[2], ref.mean(0), 3e-3, "mean from 128x128 partials")
    check(outs[3], torch.rsqrt(ref.var(0, unbiased=False) + 1e-5), 3e-3, "invstd from 128x128 partials")


G8 = ("8w256x256", "8w256x128", "8w128x256", "8w128x128")


@pytest.mark.parametrize("variant", G8 + ("64x64k2",))      # (+ the 4-wave tile with the K-steps split over two wave groups)
@pytest.mark.parametrize("case", [
    dict(B=2, H=24, W=24, C=64, N=256, k=1),          # M 1152 = 4.5 row tiles: M tail; one K-tile: prologue == whole pipeline
    dict(B=3, H=20, W=20, C=128, N=320, k=3),         # 3x3 with padding, tap changes every 2 K-tiles, N tail (320 = 256 + 64)
    dict(B=1, H=30, W=30, C=192, N=136, k=3),         # K = 27 K-tiles (odd: both ring parities end the loop), N tail < 32
    dict(B=8, H=52, W=52, C=512, N=512, k=3),         # benchmark shape M 21632 N 512 K 4608
    dict(B=8, H=104, W=104, C=256, N=128, k=1),       # M 86528 N 128 K 256
    dict(B=2, H=16, W=16, C=64, N=512, k=3, stride=2),  # strided gather
])
def test_conv_gemm_8wave_tiles(variant, case):
    """the 8-wave ping-pong tiles (csrc/gemm8.hip), each forced by name: against conv2d in fp32, against the 4-wave tile
    bit for bit (every variant adds the K-tiles in the same order), BatchNorm partials, and run-to-run identity (the
    schedule's RAW / WAR distances are by construction; a violated one shows as rare differing tiles)"""
    B, H, W, C_, N, k = case["B"], case["H"], case["W"], case["C"], case["N"], case["k"]
    stride = case.get("stride", 1)
    pad = k // 2
    x = rnd(B, H, W, C_).to(BF).float()
    w = (rnd(N, C_, k, k, seed=1) / math.sqrt(C_ * k * k)).to(BF).float()
    g = Geom(B, H, W, C_, k, k, stride, pad)
    xd, wd = bf(x), bf(pack_F(w))
    out = torch.empty(g.M, N, dtype=BF, device=DEV)
    st = ops.conv_gemm(xd, wd, g, N, out=out, stats=True, variant=variant)
    ref = conv_ref(x, w, stride, pad)
    check(out, ref, 6e-3, "%s %s" % (variant, case))
    base = torch.empty(g.M, N, dtype=BF, device=DEV)
    ops.conv_gemm(xd, wd, g, N, out=base, variant="128x128")
    if variant == "64x64k2":       # even K-steps + odd K-steps: another summation order than the other tiles
        check(out, base, 6e-3, "64x64k2 against the 128x128 tile")
    else:
        assert torch.equal(out, base), "%s differs from the 128x128 tile" % variant
    rows = st.rows_per_part
    assert rows == (32 if variant == "64x64k2" else 64 if variant in ("8w128x256", "8w128x128") else 128)
    y = ref                                            # (statistics are taken from the fp32 accumulators)
    for part in (0, (g.M - 1) // rows):
        blk = y[part * rows:(part + 1) * rows]
        check(st[0][part], blk.sum(0), 3e-3, "part sum")
        check(st[1][part], ((blk - blk.mean(0)) ** 2).sum(0), 6e-3, "part M2")
    for _ in range(6):
        again = torch.empty(g.M, N, dtype=BF, device=DEV)
        ops.conv_gemm(xd, wd, g, N, out=again, variant=variant)
        assert torch.equal(out, again), "%s is not reproducible" % variant


@pytest.mark.parametrize("variant", G8 + ("64x64k2",))
def test_conv_gemm_8wave_epilogues(variant):
    """general epilogue (bias, QuickGELU, fp32 residual / output, dropout) and the residual of the lean one on the 8-wave tiles"""
    M, K, N = 700, 320, 328                           # C = 320 = 5 K-tiles; N tail
    x = rnd(M, K).to(BF).float()
    w = (rnd(N, K, seed=1) / math.sqrt(K)).to(BF).float()
    bias, res32 = rnd(N, seed=2), rnd(M, N, seed=3)
    resbf = rnd(M, N, seed=4).to(BF).float()
    g = Geom.linear(M, K)
    base = x @ w.t()
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.conv_gemm(bf(x), bf(w), g, N, bias=bias.to(DEV), act=2, resid=res32.to(DEV), out=out, variant=variant)
    b2 = base + bias
    check(out, b2 * torch.sigmoid(1.702 * b2) + res32, 3e-3, "quickgelu+resid f32")
    wide = torch.zeros(M, N + 64, dtype=BF, device=DEV)
    ops.conv_gemm(bf(x), bf(w), g, N, resid=bf(resbf), out=wide, ldc=N + 64, c_coff=64, variant=variant)
    check(wide[:, 64:], base + resbf, 6e-3, "lean + bf16 residual into a slice")
    assert float(wide[:, :64].float().abs().max()) == 0.0
    out2 = torch.empty(M, N, dtype=BF, device=DEV)
    ops.conv_gemm(bf(x), bf(w), g, N, bias=bias.to(DEV), act=3, resid=bf(resbf), out=out2, variant=variant)
    check(out2, torch.relu(b2 + resbf), 6e-3, "EPI 2: bias + residual + relu")
    p, seed, stream = 0.1, 99, 3
    out3 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.conv_gemm(bf(x), bf(w), g, N, bias=bias.to(DEV), out=out3, drop=Drop(p, seed, stream), variant=variant)
    km = keep_mask(seed, stream, (M, N), p).cpu()
    check(out3, b2 * km / (1 - p), 3e-3, "dropout epilogue")


@pytest.mark.parametrize("M,K,N", [(136, 512, 1536), (136, 2048, 512), (136, 1536, 512), (136, 512, 2048), (144, 320, 40), (24, 512, 520)])
def test_skinny_split_k(M, K, N):
    """the split-K skinny kernels (text encoder: 136 rows): K slices over blocks, slabs added in slice order by the finishing
    launch, which runs the general epilogue - bias / QuickGELU / fp32 residual + output, bf16 output with ReLU, BatchNorm
    partials of 16 rows, the head-split transposed copy; reproducible run to run"""
    x = rnd(M, K).to(BF).float()
    w = (rnd(N, K, seed=1) / math.sqrt(K)).to(BF).float()
    bias, res32 = rnd(N, seed=2), rnd(M, N, seed=3)
    g = Geom.linear(M, K)
    base = x @ w.t() + bias
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.conv_gemm(bf(x), bf(w), g, N, bias=bias.to(DEV), act=2, resid=res32.to(DEV), out=out, variant="skinny9s")
    check(out, base * torch.sigmoid(1.702 * base) + res32, 3e-3, "quickgelu + fp32 residual")
    again = torch.empty_like(out)
    ops.conv_gemm(bf(x), bf(w), g, N, bias=bias.to(DEV), act=2, resid=res32.to(DEV), out=again, variant="skinny9s")
    assert torch.equal(out, again)
    o2 = torch.empty(M, N, dtype=BF, device=DEV)
    st = ops.conv_gemm(bf(x), bf(w), g, N, bias=bias.to(DEV), act=1, out=o2, stats=True, variant="skinny9s")
    ref = torch.relu(base)
    check(o2, ref, 6e-3, "relu bf16")
    assert st.rows_per_part == 16
    check(st[0][0], ref[:16].sum(0), 3e-3, "part sum")
    auto = torch.empty(M, N, dtype=BF, device=DEV)
    ops.conv_gemm(bf(x), bf(w), g, N, bias=bias.to(DEV), act=1, out=auto)             # the automatic choice for these shapes
    check(auto, ref, 6e-3, "automatic variant")
    if M == 136 and N % 512 == 0:                                                     # head-split transposed copy (qkv projection)
        L, Bn, E = 17, 8, 512
        Lpad, secs = ops.pad32(L), N // E
        T = torch.zeros(secs, Bn * (E // 64) * 64, Lpad, dtype=BF, device=DEV)
        o3 = torch.empty(M, N, dtype=BF, device=DEV)
        ops.conv_gemm(bf(x), bf(w), g, N, bias=bias.to(DEV), out=o3, outT=T, T_L=L, T_Lpad=Lpad, T_E=E, T_sec_stride=T[0].numel(),
                      variant="skinny9s")
        y = o3.float().cpu().view(Bn, L, secs, E // 64, 64)
        want = y.permute(2, 0, 3, 4, 1).reshape(secs, Bn * (E // 64) * 64, L)
        assert torch.equal(T[:, :, :L].float().cpu(), want)


def test_conv_gemm_variant_refused_when_not_applicable():
    x, w = bf(rnd(64, 72)), bf(rnd(64, 72, seed=1))
    out = torch.empty(64, 64, dtype=BF, device=DEV)
    with pytest.raises(ValueError):
        ops.conv_gemm(x, w, Geom.linear(64, 72), 64, out=out, variant="8w256x256")      # C % 64 != 0
    with pytest.raises(ValueError):
        ops.conv_gemm(bf(rnd(300, 64)), bf(rnd(64, 64)), Geom.linear(300, 64), 64, out=torch.empty(300, 64, dtype=BF, device=DEV),
                      variant="skinny9")                                                # M > 144


@pytest.mark.parametrize("L,B", [(24, 3), (17, 2)])
def test_conv_gemm_transposed_store(L, B):
    """head-split transposed copy written by the epilogue: outT[sec][(b*H+h)*64+d][l]"""
    E, secs = 128, 3
    M, K, N = B * L, 64, E * secs
    x = rnd(M, K).to(BF).float()
    w = (rnd(N, K, seed=1) / math.sqrt(K)).to(BF).float()
    Lpad = ops.pad32(L)
    Hh = E // 64
    outT = torch.zeros(secs, B * Hh * 64, Lpad, dtype=BF, device=DEV)
    out = torch.empty(M, N, dtype=BF, device=DEV)
    ops.conv_gemm(bf(x), bf(w), Geom.linear(M, K), N, out=out, outT=outT, T_L=L, T_Lpad=Lpad, T_E=E,
                  T_sec_stride=B * Hh * 64 * Lpad)
    y = out.float().cpu()                                    # [B*L, secs*E]
    ref = y.view(B, L, secs, Hh, 64).permute(2, 0, 3, 4, 1)   # [secs, B, Hh, 64, L]
    got = outT.float().cpu().view(secs, B, Hh, 64, Lpad)
    assert torch.equal(got[..., :L], ref.contiguous())
    assert float(got[..., L:].abs().max()) == 0.0


@pytest.mark.parametrize("L,B,secs", [(676, 2, 2), (169, 3, 1), (17, 16, 3), (900, 1, 1)])
def test_conv_gemm_general_epilogue_interior_tiles(L, B, secs):
    """the general epilogue's fast form on interior wave tiles (gemm_epilogue_fast32_gen): bias + head-split transposed copy for
    token counts with / without a multiple of 4 (packed 8-byte / 2-byte stores, batch wrap inside a wave tile), fp32 residual +
    fp32 output, QuickGELU, fp32 output alone - against torch and against the transposed layout's definition"""
    E = 128
    M, K, N = B * L, 256, E * secs
    x = rnd(M, K).to(BF).float()
    w = (rnd(N, K, seed=1) / math.sqrt(K)).to(BF).float()
    bias = rnd(N, seed=2)
    Lpad, Hh = ops.pad32(L), E // 64
    outT = torch.zeros(secs, B * Hh * 64, Lpad, dtype=BF, device=DEV)
    out = torch.empty(M, N, dtype=BF, device=DEV)
    for variant in ("64x64", "128x128"):
        outT.zero_()
        ops.conv_gemm(bf(x), bf(w), Geom.linear(M, K), N, bias=bias.to(DEV), out=out, outT=outT, T_L=L, T_Lpad=Lpad, T_E=E,
                      T_sec_stride=B * Hh * 64 * Lpad, variant=variant)
        check(out, x @ w.t() + bias, 6e-3, "bias + transposed copy (%s)" % variant)
        y = out.float().cpu()
        ref = y.view(B, L, secs, Hh, 64).permute(2, 0, 3, 4, 1)
        got = outT.float().cpu().view(secs, B, Hh, 64, Lpad)
        assert torch.equal(got[..., :L], ref.contiguous()), variant
        assert float(got[..., L:].abs().max()) == 0.0
        res32 = rnd(M, N, seed=3)
        o32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
        ops.conv_gemm(bf(x), bf(w), Geom.linear(M, K), N, bias=bias.to(DEV), act=2, resid=res32.to(DEV), out=o32, variant=variant)
        base = x @ w.t() + bias
        check(o32, base * torch.sigmoid(1.702 * base) + res32, 3e-3, "quickgelu + fp32 stream (%s)" % variant)
        ops.conv_gemm(bf(x), bf(w), Geom.linear(M, K), N, bias=bias.to(DEV), out=o32, variant=variant)
        check(o32, base, 3e-3, "fp32 out (%s)" % variant)


def _group_case(i, B, H, W, C_, N, k, epi):
    """one problem of a group test: (kwargs for ops.conv_gemm, reference output or None)"""
    x = rnd(B, H, W, C_, seed=10 + i).to(BF).float()
    w = (rnd(N, C_, k, k, seed=20 + i) / math.sqrt(C_ * k * k)).to(BF).float()
    g = Geom(B, H, W, C_, k, k, 1, k // 2)
    kw = dict(A=bf(x), Wt=bf(pack_F(w)), g=g, N=N)
    ref = conv_ref(x, w, 1, k // 2)
    if epi == "bias":
        b = rnd(N, seed=30 + i)
        kw["bias"] = b.to(DEV)
        ref = ref + b
    elif epi == "resid":
        r = rnd(g.M, N, seed=40 + i).to(BF).float()
        kw["resid"] = bf(r)
        ref = ref + r
    return kw, ref


@pytest.mark.parametrize("variant", ["64x64", "64x128", "128x64", "128x128", "8w128x128", -1])
@pytest.mark.parametrize("epi", ["lean", "bias", "resid", "stats"])
def test_conv_gemm_group_matches_single_launches(variant, epi):
    """cris_conv_gemm_group_launch: several independent problems of different sizes in one launch - outputs (and BatchNorm
    partials) bit-identical to launching each problem alone with the same tile, and right against torch"""
    shapes = [(2, 26, 26, 64, 128, 3), (1, 13, 13, 128, 256, 1), (3, 10, 12, 64, 192, 3), (2, 7, 9, 192, 72, 1), (8, 1, 1, 128, 128, 1)]
    single, grouped, refs = [], [], []
    q = ops.GemmQueue()
    for i, sh in enumerate(shapes):
        kw, ref = _group_case(i, *sh, epi)
        a, wt, g, N = kw.pop("A"), kw.pop("Wt"), kw.pop("g"), kw.pop("N")
        o1 = torch.full((g.M, N), float("nan"), dtype=BF, device=DEV)
        o2 = torch.full((g.M, N), float("nan"), dtype=BF, device=DEV)
        s1 = ops.conv_gemm(a, wt, g, N, out=o1, stats=(epi == "stats"), variant=variant, **kw)
        s2 = ops.conv_gemm(a, wt, g, N, out=o2, stats=(epi == "stats"), variant=variant, queue=q, **kw)
        single.append((o1, s1))
        grouped.append((o2, s2))
        refs.append(ref)
    assert len(q) == len(shapes)
    q.flush()
    assert len(q) == 0
    for (o1, s1), (o2, s2), ref in zip(single, grouped, refs):
        check(o2, ref, 6e-3, "grouped launch vs torch")
        assert torch.equal(o1, o2)
        if s1 is not None:
            assert s1.rows_per_part == s2.rows_per_part and torch.equal(s1.t[:, :s1.nparts], s2.t[:, :s2.nparts])


@pytest.mark.parametrize("variant", ["64x64", "64x128", "128x128", "8w128x128", -1])
@pytest.mark.parametrize("case", [dict(B=2, H=26, W=26, C=128, N=64, k=3), dict(B=3, H=13, W=11, C=64, N=256, k=1),
                                  dict(B=8, H=52, W=52, C=128, N=128, k=3)])
def test_conv_gemm_bn_backward_partials(variant, case):
    """the input-gradient GEMM's epilogue also produces the BatchNorm-backward partial sums of the layer whose output gradient it
    writes (cris_conv_gemm_params.bnr_y): summed over the row blocks they equal sum g and sum g * xhat of the STORED gradient
    with g = dz where scale * y + shift > 0 - on interior and edge tiles (M 1352 / 429 / 21632) - and the output is unchanged"""
    B, H, W, C_, N, k = case["B"], case["H"], case["W"], case["C"], case["N"], case["k"]
    x = rnd(B, H, W, C_).to(BF).float()
    w = (rnd(N, C_, k, k, seed=1) / math.sqrt(C_ * k * k)).to(BF).float()
    g = Geom(B, H, W, C_, k, k, 1, k // 2)
    y = rnd(g.M, N, seed=2).to(BF)
    mean, invstd = rnd(N, seed=3) * 0.1, 1.0 + 0.2 * rnd(N, seed=4).abs()
    scale, shift = 1.0 + 0.3 * rnd(N, seed=5), 0.2 * rnd(N, seed=6)
    dev = lambda t: t.to(DEV).contiguous()
    out = torch.empty(g.M, N, dtype=BF, device=DEV)
    plain = torch.empty(g.M, N, dtype=BF, device=DEV)
    ops.conv_gemm(bf(x), bf(pack_F(w)), g, N, out=plain, variant=variant)
    parts = ops.conv_gemm(bf(x), bf(pack_F(w)), g, N, out=out, variant=variant,
                          bnr=dict(y=dev(y), ldy=N, coff=0, mean=dev(mean), invstd=dev(invstd), scale=dev(scale), shift=dev(shift)))
    assert torch.equal(out, plain)
    if parts is None:                     # a list of more than BNR_MAX_PARTS row blocks: the GEMM ran plainly, the caller reduces as before
        assert g.M > 32 * ops.BNR_MAX_PARTS or variant == -1
        return
    assert isinstance(parts, ops.BnrParts) and parts.t.shape == (parts.nparts, 2 * N)
    dz, yf = out.float().cpu(), y.float()
    gg = torch.where(yf * scale + shift > 0, dz, torch.zeros_like(dz))
    got = parts.t.double().sum(0).cpu()
    ref0, ref1 = gg.double().sum(0), (gg.double() * ((yf.double() - mean.double()) * invstd.double())).sum(0)
    tol = 2e-4 * float(gg.abs().double().sum(0).max())
    assert float((got[:N] - ref0).abs().max()) <= tol and float((got[N:] - ref1).abs().max()) <= 2 * tol, (got[:4], ref0[:4])


def test_conv_gemm_group_chunks_and_mixed_keys():
    """more problems than CRIS_GEMM_GROUP_MAX, two epilogue kinds and a skinny problem in one queue: several launches, same results"""
    q = ops.GemmQueue()
    outs = []
    for i in range(15):
        kw, ref = _group_case(i, 1, 9 + i, 8, 64, 64 + 8 * (i % 3), 1, "bias" if i % 2 else "lean")
        a, wt, g, N = kw.pop("A"), kw.pop("Wt"), kw.pop("g"), kw.pop("N")
        o = torch.full((g.M, N), float("nan"), dtype=BF, device=DEV)
        ops.conv_gemm(a, wt, g, N, out=o, queue=q, **kw)
        outs.append((o, ref))
    kw, ref = _group_case(99, 8, 1, 1, 1024, 520, 1, "bias")           # M = 8: the skinny kernel, launched alone at the flush
    a, wt, g, N = kw.pop("A"), kw.pop("Wt"), kw.pop("g"), kw.pop("N")
    o = torch.full((g.M, N), float("nan"), dtype=BF, device=DEV)
    ops.conv_gemm(a, wt, g, N, out=o, queue=q, **kw)
    outs.append((o, ref))
    q.flush()
    for o, ref in outs:
        check(o, ref, 6e-3, "mixed queue")


def test_conv_gemm_auto_variant_without_workspace():
    """a C-ABI caller that does not know the `ws` field (NULL) still gets the text-encoder shapes computed: the automatic choice
    falls back from the split-K skinny kernel to the single-pass one"""
    M, K, N = 136, 2048, 512
    x = rnd(M, K).to(BF).float()
    w = (rnd(N, K, seed=1) / math.sqrt(K)).to(BF).float()
    p = hip.ConvGemmParams()
    xa, wa = bf(x), bf(w)
    out = torch.empty(M, N, dtype=BF, device=DEV)
    p.A, p.Wt, p.out = xa.data_ptr(), wa.data_ptr(), out.data_ptr()
    p.lda, p.ldb, p.ldc = K, K, N
    p.Bn, p.H, p.W, p.C, p.OH, p.OW, p.KH, p.KW, p.stride, p.pad = M, 1, 1, K, 1, 1, 1, 1, 1, 0
    p.M, p.N, p.K = M, N, K
    import ctypes
    hip.call("cris_conv_gemm", ctypes.byref(p), torch.cuda.current_stream().cuda_stream)
    check(out, x @ w.t(), 6e-3, "auto variant, ws = NULL")


WGRAD_CASES = [
    dict(B=2, H=12, W=12, C=64, N=128, k=1),
    dict(B=2, H=12, W=12, C=64, N=64, k=3),
    dict(B=1, H=20, W=20, C=32, N=32, k=3),
    dict(B=2, H=9, W=9, C=136, N=72, k=3, C_real=130),
    dict(B=8, H=1, W=1, C=1024, N=2305, k=1, tile=256),
    dict(B=1500, H=1, W=1, C=40, N=24, k=1),
    dict(B=2, H=26, W=26, C=128, N=256, k=3),          # M=1352: many tiles, long pixel loop with image-border carries
    dict(B=2, H=26, W=26, C=128, N=256, k=3, tile=256),    # the same on the 8-wave 256x256 tile (forced: the library picks it
    dict(B=3, H=10, W=10, C=72, N=200, k=3, tile=256),     # only for M >= 16384, K >= 4096); ragged n / k tails (N 200, K 648)
    dict(B=8, H=52, W=52, C=128, N=256, k=3, tile=256),    # benchmark shape M 21632 / N 256 / K 1152, split pixel range
    dict(B=6, H=52, W=52, C=512, N=256, k=3),          # M 16224 / K 4608: the library's own choice is the 8-wave tile
]


def _wgrad_problem(case):
    B, H, W, C_, N, k = case["B"], case["H"], case["W"], case["C"], case["N"], case["k"]
    C_real = case.get("C_real", C_)
    pad = k // 2
    x = rnd(B, H, W, C_).to(BF).float()
    if C_real < C_:
        x[..., C_real:] = 0
    g = Geom(B, H, W, C_, k, k, 1, pad)
    Nld = ops.pad8(N)
    dy = torch.zeros(g.M, Nld)
    dy[:, :N] = rnd(g.M, N, seed=5)
    dy = dy.to(BF).float()
    # reference: autograd of conv2d wrt weight
    wt = torch.zeros(N, C_real, k, k, requires_grad=True)
    y = F.conv2d(x[..., :C_real].permute(0, 3, 1, 2), wt, padding=pad)
    y.backward(dy[:, :N].reshape(B, g.OH, g.OW, N).permute(0, 3, 1, 2))
    return x, dy, g, wt.grad, C_real


def _wgrad_check(case, dWg, dbias, dy, ref, C_real, what):
    C_, N, k = case["C"], case["N"], case["k"]
    check(dbias, dy[:, :N].sum(0), 1e-5, "bias gradient fused into wgrad %s %s" % (case, what))
    dW = dWg.view(N, k * k, C_)[:, :, :C_real].permute(0, 2, 1).reshape(N, C_real, k, k)
    check(dW, ref, 3e-3, "wgrad %s %s" % (case, what))
    assert float(dWg.view(N, k * k, C_)[:, :, C_real:].abs().max() if C_real < C_ else 0.0) == 0.0


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad(case):
    """one problem per launch: unsplit (plain stores) and split pixel ranges (workspace slabs + ordered reduction); the
    result is deterministic (no atomics): a second run of the same launch is bit-identical."""
    x, dy, g, ref, C_real = _wgrad_problem(case)
    N, k, C_ = case["N"], case["k"], case["C"]
    outs = {}
    for splits in (None, 1, 3, 3):
        dWg = torch.full((N, k * k * C_), float("nan"), device=DEV)      # GEMM layout [n][tap][c]; every element is overwritten
        dbias = torch.full((N,), float("nan"), device=DEV)
        ops.conv_wgrad(bf(dy), bf(x), g, N, dWg, splits=splits, dbias=dbias, tile=case.get("tile", 0))
        _wgrad_check(case, dWg, dbias, dy, ref, C_real, "splits=%s" % splits)
        if splits in outs:
            assert torch.equal(outs[splits][0], dWg) and torch.equal(outs[splits][1], dbias), "split reduction is not deterministic"
        outs[splits] = (dWg, dbias)


def test_conv_wgrad_tile_choice():
    """which problems the launchers put on the 8-wave 256x256 tile (cris_conv_wgrad_tile): long, wide reductions
    (M >= 16384, K >= 4096, N >= 192), or whatever params.tile asks for"""
    import ctypes
    from cris.pytorch_amd import hip
    got = []
    for M, N, K, tile in [(21632, 512, 4608, 0), (21632, 256, 1152, 0), (5408, 512, 4608, 0), (21632, 128, 4608, 0),
                          (86528, 256, 4608, 0), (300, 200, 648, 256), (21632, 512, 4608, 128)]:
        p = hip.WgradParams()
        p.M, p.N, p.K, p.tile = M, N, K, tile
        got.append(hip.load().cris_conv_wgrad_tile(ctypes.byref(p)))
    assert got == [256, 128, 128, 128, 256, 256, 128], got


def test_conv_wgrad_group():
    """the queued form: all cases (twice over, so that the queue has to cut the list into more than one launch of
    CRIS_WGRAD_GROUP_MAX problems) in grouped launches; each result equals its own single-problem launch bit for bit."""
    from cris.pytorch_amd import hip
    q = ops.WgradQueue()
    jobs = []
    cases = WGRAD_CASES * 4
    assert len(cases) > hip.WGRAD_GROUP_MAX
    for case in cases:
        x, dy, g, ref, C_real = _wgrad_problem(case)
        N, k, C_ = case["N"], case["k"], case["C"]
        dWg = torch.full((N, k * k * C_), float("nan"), device=DEV)
        dbias = torch.full((N,), float("nan"), device=DEV)
        dyb, xb = bf(dy), bf(x)
        ops.conv_wgrad(dyb, xb, g, N, dWg, dbias=dbias, queue=q, tile=case.get("tile", 0))
        jobs.append((case, dWg, dbias, dy, ref, C_real, dyb, xb, g))
    assert q.items, "nothing was queued"
    q.flush()
    assert not q.items
    for case, dWg, dbias, dy, ref, C_real, dyb, xb, g in jobs:
        _wgrad_check(case, dWg, dbias, dy, ref, C_real, "grouped")
        one = torch.empty_like(dWg)
        ob = torch.empty_like(dbias)
        # (problems of more than 8192 pixel rows are not queued: they were launched at once with their automatic split)
        ops.conv_wgrad(dyb, xb, g, case["N"], one, splits=1 if g.M <= 8192 else None, dbias=ob, tile=case.get("tile", 0))
        assert torch.equal(one, dWg) and torch.equal(ob, dbias)


def test_pack_weights():
    tab = ops.PackTable()
    w3 = rnd(24, 20, 3, 3).to(DEV)               # conv: Cin 20 -> Cpad 24
    wl = rnd(40, 72, seed=1).to(DEV)             # linear [N, K]
    wt = rnd(64, 48, seed=2).to(DEV)             # used as x @ wt (text_projection): src [Cin=64][N=48]
    f3, d3 = tab.add(w3.view(24, 20, 9), 24, 20, 9, Cpad=24)
    fl, dl = tab.add(wl.view(40, 72, 1), 40, 72, 1)
    ft, dt = tab.add(wt, 48, 64, 1, src_transposed=True)
    tab.run()
    torch.cuda.synchronize()
    w3b = w3.to(BF).float().cpu()
    ref_f3 = torch.zeros(24, 9, 24)
    ref_f3[:, :, :20] = w3b.permute(0, 2, 3, 1).reshape(24, 9, 20)
    assert torch.equal(f3.float().cpu().view(24, 9, 24), ref_f3)
    ref_d3 = w3b.flip(2, 3).permute(1, 2, 3, 0).reshape(20, 9, 24)      # [c][flipped tap][n]
    assert torch.equal(d3.float().cpu().view(20, 9, 24), ref_d3)
    wlb = wl.to(BF).float().cpu()
    assert torch.equal(fl.float().cpu(), wlb)
    assert torch.equal(dl.float().cpu(), wlb.t().contiguous())
    wtb = wt.to(BF).float().cpu()
    assert torch.equal(ft.float().cpu(), wtb.t().contiguous())
    assert torch.equal(dt.float().cpu(), wtb)


def test_pack_row_stride_skew(monkeypatch):
    """round 6 (opt-in, CRIS_PACK_SKEW=1): packs whose rows would be a multiple of 256 B apart get one more 128-byte line per row
    (ops.pack_row_stride: the rows then walk through all L2 channels).  The values sit where the GEMMs look for them (ldb =
    shape[1]), the padding stays zero, a convolution / its input gradient through the skewed packs equal torch's, and the packs
    the fused Adam writes equal cris_pack_weights' (same row strides)."""
    monkeypatch.setattr(ops, "PACK_SKEW", True)
    assert ops.pack_row_stride(4608) == 4672 and ops.pack_row_stride(2304) == 2368 and ops.pack_row_stride(576) == 576
    assert ops.pack_row_stride(128) == 192 and ops.pack_row_stride(64) == 64 and ops.pack_row_stride(216) == 216
    B, H, W, C_, N = 2, 12, 12, 128, 256                        # 3x3: K = 1152 = 9 lines... x 2 B = 18 lines (even) -> skewed
    w = (rnd(N, C_, 3, 3) / math.sqrt(C_ * 9)).to(BF).float()
    wl = rnd(96, 256, seed=1).to(BF).float()                    # linear, K = 256
    tab = ops.PackTable()
    f3, d3 = tab.add(w.to(DEV).view(N, C_, 9), N, C_, 9)
    fl, dl = tab.add(wl.to(DEV).view(96, 256, 1), 96, 256, 1)
    tab.run()
    torch.cuda.synchronize()
    assert f3.shape == (N, 9 * C_ + 64) and d3.shape == (C_, 9 * N + 64) and fl.shape == (96, 256 + 64) and dl.shape == (256, 96)
    assert torch.equal(f3[:, :9 * C_].float().cpu().view(N, 9, C_), w.permute(0, 2, 3, 1).reshape(N, 9, C_))
    assert torch.equal(d3[:, :9 * N].float().cpu().view(C_, 9, N), w.flip(2, 3).permute(1, 2, 3, 0).reshape(C_, 9, N))
    assert torch.equal(fl[:, :256].float().cpu(), wl) and torch.equal(dl.float().cpu(), wl.t().contiguous())
    assert float(f3[:, 9 * C_:].abs().max()) == 0.0 and float(d3[:, 9 * N:].abs().max()) == 0.0 and float(fl[:, 256:].abs().max()) == 0.0
    x = rnd(B, H, W, C_, seed=2).to(BF).float()
    dy = rnd(B, H, W, N, seed=3).to(BF).float()
    y = torch.empty(B * H * W, N, dtype=BF, device=DEV)
    ops.conv_gemm(bf(x), f3, Geom(B, H, W, C_, 3, 3, 1, 1), N, out=y)
    xt = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    yr = F.conv2d(xt, w, padding=1)
    check(y, yr.permute(0, 2, 3, 1).reshape(-1, N), 6e-3, "forward through the skewed F pack")
    dx = torch.empty(B * H * W, C_, dtype=BF, device=DEV)
    ops.conv_gemm(bf(dy), d3, Geom(B, H, W, N, 3, 3, 1, 1), C_, out=dx)
    yr.backward(dy.permute(0, 3, 1, 2))
    check(dx, xt.grad.permute(0, 2, 3, 1).reshape(-1, C_), 6e-3, "dgrad through the skewed D pack")
    # the fused Adam writes the same packs
    params = [w.to(DEV).clone(), wl.to(DEV).clone()]
    tab2 = ops.PackTable()
    tab2.add(params[0].view(N, C_, 9), N, C_, 9)
    tab2.add(params[1].view(96, 256, 1), 96, 256, 1)
    g3 = rnd(N, C_, 3, 3, seed=5)
    grads = [g3.permute(0, 2, 3, 1).reshape(N, 9 * C_).contiguous().to(DEV), rnd(96, 256, seed=6).to(DEV)]
    adam = ops.AdamTable(params, grads, [1e-3] * 2, layouts=[(N, C_, 9, C_), None], packs=tab2.info)
    adam.step()
    got = [(f.clone(), d.clone()) for f, d, *_ in tab2.info]
    tab2.run()
    torch.cuda.synchronize()
    for (f, d, *_), (gf, gd) in zip(tab2.info, got):
        assert torch.equal(f, gf) and torch.equal(d, gd)


def test_dgrad_via_packed_weights():
    """input gradient of a 3x3 conv = the same implicit GEMM over dY with the D-layout pack"""
    B, H, W, C_, N = 2, 10, 10, 32, 48
    w = (rnd(N, C_, 3, 3) / math.sqrt(C_ * 9)).to(BF).float()
    dy = rnd(B, H, W, N, seed=3).to(BF).float()
    tab = ops.PackTable()
    _, wd = tab.add(w.to(DEV).view(N, C_, 9), N, C_, 9, want_F=False)
    tab.run()
    dx = torch.empty(B * H * W, C_, dtype=BF, device=DEV)
    ops.conv_gemm(bf(dy), wd, Geom(B, H, W, N, 3, 3, 1, 1), C_, out=dx)
    x = torch.zeros(B, C_, H, W, requires_grad=True)
    F.conv2d(x, w, padding=1).backward(dy.permute(0, 3, 1, 2))
    check(dx, x.grad.permute(0, 2, 3, 1).reshape(-1, C_), 6e-3, "dgrad")


def test_bn_stats_robust_to_large_mean():
    """|mean| >> std: E[x^2]-E[x]^2 in fp32 would lose the variance; the block-centred partials do not."""
    M, C_ = 4000, 16
    y = (rnd(M, C_) * 0.01 + 100.0).to(BF).float()         # bf16 grid near 100 has spacing 0.5: values are 99.5/100/100.5
    y = y + 0.0
    st = ops.colstats(bf(y), M, C_, 32, DEV)
    outs = [torch.empty(C_, device=DEV) for _ in range(4)]
    ops.bn_finalize(st, M, M, torch.ones(C_, device=DEV), torch.zeros(C_, device=DEV), None, None, 0.1, 1e-5, C_, *outs)
    check(outs[2], y.double().mean(0), 1e-6, "mean")
    check(outs[3], torch.rsqrt(y.double().var(0, unbiased=False) + 1e-5), 1e-4, "invstd")


@pytest.mark.parametrize("M,rows", [(40000, 32), (5000, 32), (86528, 128), (346112, 128), (3000, 32)])
def test_bn_finalize_long_lists(M, rows):
    """lists of 129 - 512 partials are merged by 64 lanes per channel inside the finalize launch (157 parts here), longer ones
    (1250, 676, 2704 parts) through a first-level merge into 64 slices, ragged last slice included; 94 parts take the 16-lane
    form.  The result must equal the direct statistics."""
    C_ = 72
    y = (rnd(M, C_) * 2.0 + 3.0).to(BF).float()
    st = ops.colstats(bf(y), M, C_, rows, DEV)
    assert st.nparts == (M + rows - 1) // rows and st.t.shape[1] == st.nparts + (64 if st.nparts > 512 else 0)
    outs = [torch.empty(C_, device=DEV) for _ in range(4)]
    ops.bn_finalize(st, M, M, torch.ones(C_, device=DEV), torch.zeros(C_, device=DEV), None, None, 0.1, 1e-5, C_, *outs)
    check(outs[2], y.double().mean(0), 1e-6, "mean")
    check(outs[3], torch.rsqrt(y.double().var(0, unbiased=False) + 1e-5), 1e-5, "invstd")


# ----------------------------------------------------------------------------------------------------
# BatchNorm
# ----------------------------------------------------------------------------------------------------
def _bn_setup(B, H, W, C_, seed=0):
    y = (rnd(B, H, W, C_, seed=seed) * 1.5 + 0.3).to(BF).float()
    gamma = rnd(C_, seed=seed + 1) * 0.2 + 1.0
    beta = rnd(C_, seed=seed + 2) * 0.1
    return y, gamma, beta


def _bn_coeffs(y2d, gamma, beta, rm=None, rv=None):
    C_ = y2d.shape[1]
    M = y2d.shape[0]
    st = ops.colstats(bf(y2d), M, C_, 32, DEV)
    outs = [torch.empty(C_, device=DEV) for _ in range(4)]
    ops.bn_finalize(st, M, M, gamma.to(DEV), beta.to(DEV), rm, rv, 0.1, 1e-5, C_, *outs)
    return outs


def test_bn_finalize_and_running_stats():
    y, gamma, beta = _bn_setup(2, 6, 6, 40)
    y2 = y.reshape(-1, 40)
    rm = torch.zeros(40, device=DEV) + 0.5
    rv = torch.ones(40, device=DEV) * 2
    scale, shift, mean, invstd = _bn_coeffs(y2, gamma, beta, rm, rv)
    m = y2.mean(0)
    v = y2.var(0, unbiased=False)
    check(mean, m, 1e-5)
    check(invstd, torch.rsqrt(v + 1e-5), 1e-4)
    check(scale, gamma * torch.rsqrt(v + 1e-5), 1e-4)
    check(shift, beta - m * gamma * torch.rsqrt(v + 1e-5), 1e-3)
    check(rm, 0.9 * 0.5 + 0.1 * m, 1e-5)
    check(rv, 0.9 * 2 + 0.1 * y2.var(0, unbiased=True), 1e-4)


@pytest.mark.parametrize("shape", [(2, 8, 8, 48), (3, 10, 10, 64), (4, 100, 100, 64), (2, 6, 6, 2048)])
@pytest.mark.parametrize("variant", ["plain", "pool", "ident", "two", "mul"])
def test_bn_apply_and_backward(variant, shape):
    """C = 48: the generic kernels (C/8 not a power of two); C = 64 / 2048: the fast apply kernels (one channel vector per thread) -
    one row per thread at M = 300 / 72, several passes with a ragged last one at M = 40000; pool / mul always take the generic ones"""
    B, H, W, C_ = shape
    y, gamma, beta = _bn_setup(B, H, W, C_)
    y2d = y.reshape(-1, C_)
    M = y2d.shape[0]
    scale, shift, mean, invstd = _bn_coeffs(y2d, gamma, beta)
    yl = y.clone().requires_grad_(True)
    gl, bl = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)

    def tbn(t, g_, b_):
        return F.batch_norm(t.permute(0, 3, 1, 2), None, None, g_, b_, True, 0.1, 1e-5).permute(0, 2, 3, 1)

    kw, bkw = {}, {}
    extra_leaves = {}
    if variant == "plain":
        ref = torch.relu(tbn(yl, gl, bl))
    elif variant == "pool":
        ref = F.avg_pool2d(torch.relu(tbn(yl, gl, bl)).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
        kw["pool"] = True
    elif variant == "ident":
        idt = rnd(B, H, W, C_, seed=9).to(BF).float()
        il = idt.clone().requires_grad_(True)
        extra_leaves["ident"] = il
        ref = torch.relu(tbn(yl, gl, bl) + il)
        kw.update(ident=bf(idt))
    elif variant == "two":
        y2, gamma2, beta2 = _bn_setup(B, H, W, C_, seed=20)
        sc2, sh2, mean2, inv2 = _bn_coeffs(y2.reshape(-1, C_), gamma2, beta2)
        y2l = y2.clone().requires_grad_(True)
        g2l, b2l = gamma2.clone().requires_grad_(True), beta2.clone().requires_grad_(True)
        extra_leaves.update(y2=y2l, g2=g2l, b2=b2l)
        ref = torch.relu(tbn(yl, gl, bl) + tbn(y2l, g2l, b2l))
        kw.update(y2=bf(y2), scale2=sc2, shift2=sh2)
    elif variant == "mul":
        mul = torch.relu(rnd(B, C_, seed=11)) + 0.1
        ml = mul.clone().requires_grad_(True)
        extra_leaves["mul"] = ml
        ref = torch.relu(tbn(yl, gl, bl)) * ml[:, None, None, :]
        kw.update(mul=mul.to(DEV))
    OHo, OWo = (H // 2, W // 2) if variant == "pool" else (H, W)
    z = torch.empty(B * OHo * OWo, C_, dtype=BF, device=DEV)
    ops.bn_apply(bf(y), scale, shift, z, B, H, W, C_, relu=True, **kw)
    check(z, ref.reshape(-1, C_), 6e-3, "bn_apply " + variant)

    # backward
    dz = rnd(B, OHo, OWo, C_, seed=30).to(BF).float()
    (ref * dz).sum().backward()
    sums = torch.zeros(4 * C_, device=DEV)
    dy = torch.empty(M, C_, dtype=BF, device=DEV)
    if variant == "pool":
        bkw.update(pool=True)
    if variant == "ident":
        bkw.update(z=z, dident=torch.empty(M, C_, dtype=BF, device=DEV))
    if variant == "two":
        bkw.update(z=z, y2=bf(y2), mean2=mean2, invstd2=inv2, scale2=sc2, dy2=torch.empty(M, C_, dtype=BF, device=DEV))
    if variant == "mul":
        bkw.update(mul=mul.to(DEV), dmul=torch.zeros(B, C_, device=DEV))
    ops.bn_bwd(bf(dz), bf(y), scale, shift, mean, invstd, sums, dy, B, H, W, C_, M, relu=True, **bkw)
    check(dy, yl.grad.reshape(-1, C_), 1.5e-2, "bn dy " + variant)
    check(sums[:C_], bl.grad, 5e-3, "dbeta " + variant)
    check(sums[C_:2 * C_], gl.grad, 5e-3, "dgamma " + variant)
    if variant == "ident":
        check(bkw["dident"], extra_leaves["ident"].grad.reshape(-1, C_), 1e-2, "dident")
    if variant == "two":
        check(bkw["dy2"], extra_leaves["y2"].grad.reshape(-1, C_), 1.5e-2, "dy2")
        check(sums[3 * C_:], extra_leaves["g2"].grad, 5e-3, "dgamma2")
    if variant == "mul":
        check(bkw["dmul"], extra_leaves["mul"].grad, 5e-3, "dmul")


# ----------------------------------------------------------------------------------------------------
# LayerNorm
# ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C_", [128, 512, 2048])
def test_layernorm_variants(C_):
    rows, T = 70, 10
    x32 = rnd(rows, C_) * 2 + 0.5
    gamma = rnd(C_, seed=1) * 0.2 + 1
    beta = rnd(C_, seed=2) * 0.1
    pos = rnd(T, C_, seed=3)
    resid = rnd(rows, C_, seed=4)
    mean = torch.empty(rows, device=DEV)
    rstd = torch.empty(rows, device=DEV)
    # (a) fp32 in -> y, ypos
    y = torch.empty(rows, C_, dtype=BF, device=DEV)
    ypos = torch.empty(rows, C_, dtype=BF, device=DEV)
    ops.ln_fwd(x32.to(DEV), gamma.to(DEV), beta.to(DEV), rows, C_, mean, rstd, y=y, ypos=ypos, pos=pos.to(DEV), pos_rows=T)
    xl = x32.clone().requires_grad_(True)
    gl, bl = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(xl, (C_,), gl, bl, 1e-5)
    check(y, ref, 6e-3, "ln y")
    check(ypos, ref + pos.repeat(rows // T, 1), 6e-3, "ln ypos")
    dy = rnd(rows, C_, seed=5).to(BF).float()
    dyp = rnd(rows, C_, seed=6).to(BF).float()
    (ref * (dy + dyp)).sum().backward()
    dx = torch.zeros(rows, C_, device=DEV) + 1.0
    dg = torch.zeros(C_, device=DEV)
    db = torch.zeros(C_, device=DEV)
    ops.ln_bwd(x32.to(DEV), gamma.to(DEV), mean, rstd, rows, C_, dx, dy=bf(dy), dypos=bf(dyp), dgamma=dg, dbeta=db, dx_accum=True)
    check(dx - 1.0, xl.grad, 2e-3, "ln dx (accum f32)")
    check(dg, gl.grad, 1e-3, "ln dgamma")
    check(db, bl.grad, 1e-3, "ln dbeta")
    # (b) bf16 pre-activation in, relu + input dropout (FFN norm), bf16 dx
    p, seed = 0.1, 77
    h = (rnd(rows, C_, seed=8) * 1.5).to(BF).float()
    km = keep_mask(seed, 4, (rows, C_), p).cpu().float()
    hl = h.clone().requires_grad_(True)
    ref2 = F.layer_norm(torch.relu(hl) * km / (1 - p), (C_,), gamma, beta, 1e-5)
    y2 = torch.empty(rows, C_, dtype=BF, device=DEV)
    ops.ln_fwd(bf(h), gamma.to(DEV), beta.to(DEV), rows, C_, mean, rstd, y=y2, in_relu=True, in_drop=Drop(p, seed, 4))
    check(y2, ref2, 6e-3, "ln(relu,dropout)")
    (ref2 * dy).sum().backward()
    dh = torch.empty(rows, C_, dtype=BF, device=DEV)
    ops.ln_bwd(bf(h), gamma.to(DEV), mean, rstd, rows, C_, dh, dy=bf(dy), in_relu=True, in_drop=Drop(p, seed, 4))
    check(dh, hl.grad, 8e-3, "ln(relu,dropout) dx")
    # (c) bf16 in -> resid + dropout(LN(x)) in fp32 (post-attention norms)
    a = rnd(rows, C_, seed=9).to(BF).float()
    km2 = keep_mask(seed, 1, (rows, C_), p).cpu().float()
    al = a.clone().requires_grad_(True)
    ref3 = resid + F.layer_norm(al, (C_,), gamma, beta, 1e-5) * km2 / (1 - p)
    o3 = torch.empty(rows, C_, device=DEV)
    ops.ln_fwd(bf(a), gamma.to(DEV), beta.to(DEV), rows, C_, mean, rstd, resid=resid.to(DEV), out_f32=o3, out_drop=Drop(p, seed, 1))
    check(o3, ref3, 2e-3, "resid + drop(LN)")
    do = rnd(rows, C_, seed=10)
    (ref3 * do).sum().backward()
    da = torch.empty(rows, C_, dtype=BF, device=DEV)
    ops.ln_bwd(bf(a), gamma.to(DEV), mean, rstd, rows, C_, da, dout_f32=do.to(DEV), out_drop=Drop(p, seed, 1))
    check(da, al.grad, 8e-3, "resid-drop LN dx")


# ----------------------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------------------
def _head_T(x, B, L, Hn, Lpad):
    """[B*L, Hn*64] -> [(b*Hn+h)*64+d][Lpad] zero padded"""
    t = x.view(B, L, Hn, 64).permute(0, 2, 3, 1).reshape(B * Hn * 64, L)
    out = torch.zeros(B * Hn * 64, Lpad, dtype=x.dtype, device=x.device)
    out[:, :L] = t
    return out.contiguous()


@pytest.mark.parametrize("case", [
    dict(B=2, Hn=2, Lq=100, Lk=100, p=0.1),                  # decoder self-attention style, dropout
    dict(B=2, Hn=2, Lq=70, Lk=17, pad=True, p=0.1),          # cross attention, key padding
    dict(B=3, Hn=2, Lq=17, Lk=17, causal=True, p=0.0),       # text
    dict(B=1, Hn=4, Lq=169, Lk=169, p=0.0),                  # attnpool
    dict(B=1, Hn=2, Lq=676, Lk=676, p=0.1),                  # decoder self-attention at 416x416 (26x26 tokens)
    dict(B=1, Hn=2, Lq=900, Lk=900, p=0.0),                  # ... at 480x480 (30x30 tokens)
    dict(B=2, Hn=2, Lq=676, Lk=17, pad=True, p=0.1),         # decoder cross attention at 416x416
    dict(B=2, Hn=1, Lq=900, Lk=22, pad=True, p=0.1),         # ... at 480x480, 22-token text
])
def test_attention_fwd_bwd(case):
    B, Hn, Lq, Lk, p = case["B"], case["Hn"], case["Lq"], case["Lk"], case["p"]
    causal = case.get("causal", False)
    E = Hn * 64
    scale = 64 ** -0.5
    q = rnd(B * Lq, E).to(BF).float()
    k = rnd(B * Lk, E, seed=1).to(BF).float()
    v = rnd(B * Lk, E, seed=2).to(BF).float()
    toks = None
    if case.get("pad"):
        toks = torch.ones(B, Lk, dtype=torch.int64)
        toks[0, 9:] = 0
        if B > 1:
            toks[1, 14:] = 0
    seed, stream = 4321, 2
    Lkp, Lqp = ops.pad32(Lk), ops.pad32(Lq)
    qd, kd, vd = bf(q), bf(k), bf(v)
    # the params struct holds raw pointers: keep every operand alive until the launches are done
    vt, kt, qt = _head_T(vd, B, Lk, Hn, Lkp), _head_T(kd, B, Lk, Hn, Lkp), _head_T(qd, B, Lq, Hn, Lqp)
    toks_d = None if toks is None else toks.to(DEV)
    prm = ops.attn_params(qd, kd, vd, vt, B, Hn, Lq, Lk, Lkp, scale, Kt=kt, Qt=qt, Lq_pad=Lqp, key_tokens=toks_d,
                          causal=causal, drop=Drop(p, seed, stream))
    O = torch.empty(B * Lq, E, dtype=BF, device=DEV)
    lse = torch.empty(B * Hn, Lq, device=DEV)
    ops.attn_fwd(prm, O, lse)
    # reference
    ql, kl, vl = [t.clone().requires_grad_(True) for t in (q, k, v)]
    qh = ql.view(B, Lq, Hn, 64).transpose(1, 2) * scale
    kh = kl.view(B, Lk, Hn, 64).transpose(1, 2)
    vh = vl.view(B, Lk, Hn, 64).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    if causal:
        s = s + torch.full((Lq, Lk), float("-inf")).triu_(1)
    if toks is not None:
        s = s.masked_fill((toks == 0)[:, None, None, :], float("-inf"))
    pr = torch.softmax(s, -1)
    check(lse, torch.logsumexp(s, -1).reshape(B * Hn, Lq), 2e-3, "lse")
    if p > 0:
        km = keep_mask(seed, stream, (B, Hn, Lq, Lk), p).cpu().float()
        pr = pr * km / (1 - p)
    o = (pr @ vh).transpose(1, 2).reshape(B * Lq, E)
    check(O, o, 1e-2, "attn fwd %s" % case)
    do = rnd(B * Lq, E, seed=7).to(BF).float()
    (o * do).sum().backward()
    dod = bf(do)
    dQ = torch.empty_like(qd)
    dK = torch.empty_like(kd)
    dV = torch.empty_like(vd)
    delta = torch.empty(B * Hn, Lq, device=DEV)
    dot = _head_T(dod, B, Lq, Hn, Lqp)
    ops.attn_bwd(prm, O, lse, dod, dot, delta, dQ, dK, dV)
    torch.cuda.synchronize()
    check(dV, vl.grad, 1.5e-2, "dV")
    check(dQ, ql.grad, 1.5e-2, "dQ")
    check(dK, kl.grad, 1.5e-2, "dK")


# ----------------------------------------------------------------------------------------------------
# elementwise
# ----------------------------------------------------------------------------------------------------
def test_stem_im2col_matches_conv():
    B, H, W = 2, 20, 28
    img = rnd(B, 3, H, W)
    w = rnd(16, 3, 3, 3, seed=1).to(BF).float()
    col = torch.empty(B * (H // 2) * (W // 2), 32, dtype=BF, device=DEV)
    ops.stem_im2col(img.to(DEV), col)
    wp = torch.zeros(16, 32)
    wp[:, :27] = w.reshape(16, 27)
    out = col.float().cpu() @ wp.t()
    ref = F.conv2d(img.to(BF).float(), w, stride=2, padding=1).permute(0, 2, 3, 1).reshape(-1, 16)
    check(out, ref, 1e-5, "stem im2col")


def test_pool_and_upsample():
    B, H, W, C_ = 2, 6, 10, 24
    x = rnd(B, H, W, C_).to(BF).float()
    y = torch.empty(B * (H // 2) * (W // 2), C_, dtype=BF, device=DEV)
    ops.avgpool2_fwd(bf(x), B, H, W, C_, y)
    check(y, F.avg_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).reshape(-1, C_), 4e-3, "avgpool")
    up = torch.empty(B * H * 2 * W * 2, C_ + 8, dtype=BF, device=DEV)
    ops.upsample2_fwd(bf(x), B, H, W, C_, up, ldy=C_ + 8, ycoff=8)
    xl = x.clone().requires_grad_(True)
    ref = F.interpolate(xl.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear").permute(0, 2, 3, 1)
    check(up[:, 8:], ref.reshape(-1, C_), 4e-3, "upsample2")
    g = rnd(B, 2 * H, 2 * W, C_, seed=3).to(BF).float()
    (ref * g).sum().backward()
    dx = torch.empty(B * H * W, C_, dtype=BF, device=DEV)
    ops.upsample2_bwd(bf(g), B, H, W, C_, dx)
    check(dx, xl.grad.reshape(-1, C_), 5e-3, "upsample2 bwd")
    gp = rnd(B, H // 2, W // 2, C_, seed=4).to(BF).float()
    dxp = torch.empty(B * H * W, C_, dtype=BF, device=DEV)
    ops.avgpool2_bwd(bf(gp), B, H, W, C_, dxp)
    refp = F.interpolate(gp.permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1) * 0.25
    check(dxp, refp.reshape(-1, C_), 4e-3, "avgpool bwd")


def test_coords_adds_casts():
    B, H, W = 2, 5, 7
    buf = torch.ones(B * H * W, 16, dtype=BF, device=DEV)
    ops.fill_coords(buf, 16, 8, 8, B, H, W)
    got = buf.float().cpu().view(B, H, W, 16)
    xr = torch.linspace(-1, 1, W).to(BF).float()
    yr = torch.linspace(-1, 1, H).to(BF).float()
    assert torch.equal(got[..., 8], xr.view(1, 1, W).expand(B, H, W))
    assert torch.equal(got[..., 9], yr.view(1, H, 1).expand(B, H, W))
    assert float(got[..., 10:].abs().max()) == 0 and float((got[..., :8] - 1).abs().max()) == 0
    a = rnd(30, 16).to(BF).float()
    b = rnd(30, 16, seed=1).to(BF).float()
    y = torch.empty(30, 16, dtype=BF, device=DEV)
    ops.add_bf16(bf(a), y, 30, 16, b=bf(b))
    check(y, a + b, 4e-3)
    tab = rnd(6, 16, seed=2)
    ops.add_rowtable(bf(a), tab.to(DEV), 6, y, 30, 16)
    check(y, a + tab.repeat(5, 1), 4e-3)
    f = rnd(1000)
    o = torch.empty(1000, dtype=BF, device=DEV)
    ops.cast_f32_bf16(f.to(DEV), o)
    assert torch.equal(o.cpu(), f.to(BF))
    acc = torch.ones(1000, device=DEV)
    ops.cast_bf16_f32(o, acc, accum=True)
    check(acc, f.to(BF).float() + 1, 1e-6)


def test_embedding_and_eot():
    B, L, D, V = 3, 9, 64, 500
    toks = torch.tensor([[400, 5, 7, 499, 0, 0, 0, 0, 0], [400, 3, 499, 0, 0, 0, 0, 0, 0], [400, 1, 2, 3, 4, 5, 6, 499, 0]])
    table = rnd(V, D)
    pos = rnd(77, D, seed=1)
    out = torch.empty(B * L, D, device=DEV)
    ops.embed_fwd(toks.to(DEV), table.to(DEV), pos.to(DEV), out)
    assert torch.equal(out.cpu(), (table[toks] + pos[:L]).reshape(B * L, D))
    dx = rnd(B * L, D, seed=2)
    dt = torch.zeros(V, D, device=DEV)
    dp = torch.zeros(77, D, device=DEV)
    live = torch.zeros(V, dtype=torch.uint8, device=DEV)
    live[17] = 1                                         # sticky: marks of earlier batches stay
    ops.embed_bwd(toks.to(DEV), dx.to(DEV), dt, dp, row_live=live)
    assert sorted(live.nonzero().flatten().tolist()) == sorted(set(toks.flatten().tolist()) | {17})
    rt = torch.zeros(V, D).index_add_(0, toks.flatten(), dx)
    check(dt, rt, 1e-6)
    check(dp[:L], dx.view(B, L, D).sum(0), 1e-6)
    x = rnd(B * L, D, seed=3).to(BF)
    rows = torch.empty(B, D, dtype=BF, device=DEV)
    idx = torch.empty(B, dtype=torch.int32, device=DEV)
    ops.eot_gather(toks.to(DEV), x.to(DEV), D, rows, idx)
    assert idx.cpu().tolist() == toks.argmax(-1).tolist() == [3, 2, 7]
    assert torch.equal(rows.cpu(), x.view(B, L, D)[torch.arange(B), toks.argmax(-1)])
    dxx = torch.zeros(B * L, D, dtype=BF, device=DEV)
    ops.eot_scatter_add(idx, rows, B, L, D, dxx)
    ref = torch.zeros(B, L, D, dtype=BF)
    ref[torch.arange(B), toks.argmax(-1)] = x.view(B, L, D)[torch.arange(B), toks.argmax(-1)]
    assert torch.equal(dxx.cpu().view(B, L, D), ref)


def test_posresize_and_rowsum():
    from cris.pytorch_amd.tables import bicubic_resize_matrix
    G, H, W, C_ = 7, 13, 13, 32
    pos = rnd(G * G + 1, C_)
    R = torch.from_numpy(bicubic_resize_matrix(G, H, W)).float()
    ref = F.interpolate(pos[1:].reshape(1, G, G, C_).permute(0, 3, 1, 2), size=(H, W), mode="bicubic", align_corners=False)
    ref = ref.flatten(2)[0].t()                                       # [HW, C]
    posr = torch.empty(H * W, C_, device=DEV)
    ops.posresize_fwd(R.to(DEV), pos.to(DEV), H * W, G, C_, posr)
    check(posr, ref, 1e-5, "bicubic pos resize")
    d = rnd(H * W, C_, seed=1)
    dpos = torch.zeros(G * G + 1, C_, device=DEV)
    ops.posresize_bwd(R.to(DEV), d.to(DEV), H * W, G, C_, dpos)
    check(dpos[1:], R.t() @ d, 1e-5)
    assert float(dpos[0].abs().max()) == 0
    dx = rnd(3 * 20, 16, seed=2).to(BF)
    out = torch.empty(20, 16, device=DEV)
    ops.batch_rowsum(dx.to(DEV), 3, 20, 16, out)
    check(out, dx.float().view(3, 20, 16).sum(0), 1e-6)


@pytest.mark.parametrize("C_", [64, 256])
def test_dynconv(C_):
    B, H, W = 3, 12, 14
    x = rnd(B, H, W, C_).to(BF).float()
    wb = rnd(B, C_ * 9 + 1, seed=1) * 0.05
    pred = torch.empty(B, H * W, device=DEV)
    ops.dynconv_fwd(bf(x), B, H, W, C_, wb.to(DEV), pred)
    xl = x.clone().requires_grad_(True)
    wl = wb.clone().requires_grad_(True)
    ref = F.conv2d(xl.permute(0, 3, 1, 2).reshape(1, B * C_, H, W), wl[:, :-1].reshape(B, C_, 3, 3), padding=1, groups=B,
                   bias=wl[:, -1])[0]
    check(pred, ref.reshape(B, H * W), 1e-4, "dynconv fwd")
    dp = rnd(B, H, W, seed=2)
    (ref * dp).sum().backward()
    dx = torch.empty(B * H * W, C_, dtype=BF, device=DEV)
    dwb = torch.zeros(B, C_ * 9 + 1, device=DEV)
    ops.dynconv_bwd(bf(x), dp.to(DEV), B, H, W, C_, wb.to(DEV), dx, dwb)
    check(dx, xl.grad.reshape(-1, C_), 5e-3, "dynconv dx")
    check(dwb, wl.grad, 1e-4, "dynconv dw")


def test_loss_mask_metric():
    B, S, O_ = 3, 64, 16
    mask = (torch.rand(B, 1, S, S, generator=torch.Generator().manual_seed(0)) > 0.5).float()
    out = torch.empty(B, 1, O_, O_, device=DEV)
    ops.mask_resize_nearest(mask.to(DEV), O_, O_, out)
    assert torch.equal(out.cpu(), F.interpolate(mask, (O_, O_), mode="nearest"))
    mask2 = (torch.rand(2, 1, 50, 70, generator=torch.Generator().manual_seed(1)) > 0.5).float()
    out2 = torch.empty(2, 1, 13, 23, device=DEV)
    ops.mask_resize_nearest(mask2.to(DEV), 13, 23, out2)
    assert torch.equal(out2.cpu(), F.interpolate(mask2, (13, 23), mode="nearest"))
    x = rnd(B, 1, O_, O_) * 3
    t = out.cpu()
    loss = torch.full((1,), float('nan'), device=DEV)          # overwritten, not accumulated
    ops.bce_fwd(x.to(DEV), out, loss)
    xl = x.clone().requires_grad_(True)
    ref = F.binary_cross_entropy_with_logits(xl, t)
    assert abs(float(loss) - float(ref)) < 1e-5
    (ref * 3.0).backward()
    dx = torch.empty(B, 1, O_, O_, device=DEV)
    ops.bce_bwd(x.to(DEV), out, torch.tensor([3.0], device=DEV), dx)
    check(dx, xl.grad, 1e-5)
    met = torch.full((2,), float('nan'), device=DEV)
    ops.train_metric(x.to(DEV), out, B, O_ * O_, met)
    o = (torch.sigmoid(x.flatten(1)) >= 0.35)
    tt = t.flatten(1).bool()
    ious = (o & tt).sum(1) / ((o | tt).sum(1) + 1e-6)
    assert abs(float(met[0]) - float(100 * ious.mean())) < 1e-3
    assert abs(float(met[1]) - float(100 * (ious > 0.5).float().mean())) < 1e-3


def test_quickgelu_castdrop_axpy():
    x = (rnd(40, 64) * 2).to(BF).float()
    y = torch.empty(40, 64, dtype=BF, device=DEV)
    ops.quickgelu_fwd(bf(x), y)
    xl = x.clone().requires_grad_(True)
    ref = xl * torch.sigmoid(1.702 * xl)
    check(y, ref, 4e-3, "quickgelu")
    g = rnd(40, 64, seed=1).to(BF).float()
    (ref * g).sum().backward()
    dx = torch.empty(40, 64, dtype=BF, device=DEV)
    ops.quickgelu_bwd(bf(x), bf(g), dx)
    check(dx, xl.grad, 5e-3, "quickgelu bwd")
    f = rnd(50, 32, seed=2)
    o = torch.empty(50, 32, dtype=BF, device=DEV)
    ops.cast_f32_bf16_drop(f.to(DEV), o, Drop(0.1, 99, 5))
    km = keep_mask(99, 5, (50, 32), 0.1).cpu()
    assert torch.equal(o.cpu(), (f * km / (1 - 0.1)).to(BF)) or relerr(o, f * km / 0.9) < 4e-3
    assert bool(((o.cpu().float() == 0) | km).all()) and bool(((o.cpu().float() != 0) | ~km | (f == 0)).all())
    a = rnd(1000, seed=3).to(DEV)
    b = rnd(1000, seed=4).to(DEV)
    ref2 = a + 0.5 * b
    ops.axpy_f32(a, b, 0.5)
    check(a, ref2, 1e-6)


def test_adam_matches_torch():
    ps = [rnd(1000), rnd(33, 7, seed=1), rnd(20000, seed=2)]
    gs = [rnd(*p.shape, seed=5) for p in ps]
    dev_p = [p.clone().to(DEV) for p in ps]
    dev_g = [g.clone().to(DEV) for g in gs]
    tab = ops.AdamTable(dev_p, dev_g, [1e-3, 1e-3, 1e-4])
    ref_p = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = torch.optim.Adam([{"params": ref_p[:2], "lr": 1e-3}, {"params": ref_p[2:], "lr": 1e-4}])
    for step in range(3):
        for rp, g in zip(ref_p, gs):
            rp.grad = g * (step + 1)
        for dg, g in zip(dev_g, gs):
            dg.copy_(g * (step + 1))
        opt.step()
        tab.step()
    for dp, rp in zip(dev_p, ref_p):
        check(dp, rp.data, 1e-6, "adam")


def test_adam_row_skip_is_bit_identical_to_dense():
    """row_live: rows of an embedding table that never had a gradient are skipped; parameters and both moments stay bit-identical
    to the dense update over steps in which new rows become live, and a non-zero weight decay switches the skipping off"""
    V, D = 3000, 48                                      # 144000 elements: 18 blocks of 8192, rows straddle block edges
    table = rnd(V, D)
    other = rnd(777, seed=1)
    for wd in (0.0, 0.01):
        runs = []
        for use_live in (False, True):
            p = [table.clone().to(DEV), other.clone().to(DEV)]
            g = [torch.zeros(V, D, device=DEV), torch.zeros(777, device=DEV)]
            live = torch.zeros(V, dtype=torch.uint8, device=DEV)
            tab = ops.AdamTable(p, g, [1e-3, 1e-3], row_live={0: live} if use_live else None)
            for step in range(4):
                rows = torch.tensor([3, 170, 171, 2999, 5 + 11 * step], device=DEV)
                g[0].zero_()
                g[0][rows] = rnd(5, D, seed=10 + step).to(DEV)
                g[1].copy_(rnd(777, seed=20 + step))
                live[rows] = 1
                tab.step(weight_decay=wd)
            runs.append((p[0].clone(), tab.m[0].clone(), tab.v[0].clone(), p[1].clone()))
        for a, b in zip(*runs):
            assert torch.equal(a, b), "wd=%g" % wd
        untouched = torch.ones(V, dtype=torch.bool)
        untouched[[3, 170, 171, 2999, 5, 16, 27, 38]] = False
        if wd == 0.0:
            assert torch.equal(runs[1][0].cpu()[untouched], table[untouched])      # (and the dense update leaves them alone too)


def test_adam_gemm_layout_gradient_and_device_step():
    """conv-weight gradient in the GEMM layout [n][tap][cpad] (what cris_conv_wgrad writes) + step count on the device"""
    N, Cin, k, Cpad = 24, 20, 3, 24
    w = rnd(N, Cin, k, k)
    g = rnd(N, Cin, k, k, seed=3)
    g_gemm = torch.zeros(N, k * k, Cpad)
    g_gemm[:, :, :Cin] = g.permute(0, 2, 3, 1).reshape(N, k * k, Cin)
    g_gemm[:, :, Cin:] = 7.0                                  # padding must never be read
    dev_p, dev_g = w.clone().to(DEV), g_gemm.reshape(N, k * k * Cpad).to(DEV)
    tab = ops.AdamTable([dev_p], [dev_g], [1e-3], layouts=[(N, Cin, k * k, Cpad)])
    rp = torch.nn.Parameter(w.clone())
    opt = torch.optim.Adam([rp], lr=1e-3)
    step_dev = torch.zeros(1, dtype=torch.int32, device=DEV)
    for step in range(3):
        rp.grad = g.clone()
        opt.step()
        step_dev += 1
        tab.step_count = 100                                  # host counter deliberately wrong: the device count must win
        tab.step(step_dev=step_dev)
    check(dev_p, rp.data, 1e-6, "adam gemm-layout")


def test_adam_refreshes_the_bf16_packs():
    """the optimizer update also rewrites the F / D operand copies of the GEMM weights (no separate packing pass): after a
    step the packs written by cris_adam_step equal, bit for bit, what cris_pack_weights makes of the updated parameters -
    3x3 conv with ragged N / Cin and a GEMM-layout gradient, a linear weight, a transposed ([in, out]) projection, next to
    an unpacked tensor in the same table; parameters agree with torch.optim.Adam."""
    N3, C3, Cp3 = 70, 66, 72
    w3 = rnd(N3, C3, 3, 3).to(DEV)
    g3 = rnd(N3, C3, 3, 3, seed=3)
    g3g = torch.zeros(N3, 9, Cp3)
    g3g[:, :, :C3] = g3.permute(0, 2, 3, 1).reshape(N3, 9, C3)
    wl, gl = rnd(100, 72, seed=1).to(DEV), rnd(100, 72, seed=4)
    wt, gt = rnd(64, 136, seed=2).to(DEV), rnd(64, 136, seed=5)          # [Cin=64][N=136], used as x @ wt
    wb, gb = rnd(300, seed=6).to(DEV), rnd(300, seed=7)                   # a bias: no packs
    tab = ops.PackTable()
    tab.add(w3.view(N3, C3, 9), N3, C3, 9, Cpad=Cp3)
    tab.add(wl.view(100, 72, 1), 100, 72, 1)
    tab.add(wt, 136, 64, 1, src_transposed=True)
    params = [w3, wl, wt, wb]
    grads = [g3g.reshape(N3, 9 * Cp3).to(DEV), gl.to(DEV), gt.to(DEV), gb.to(DEV)]
    adam = ops.AdamTable(params, grads, [1e-3] * 4, layouts=[(N3, C3, 9, Cp3), None, None, None], packs=tab.info + [None])
    assert adam.refreshes_packs
    ref = [torch.nn.Parameter(t.detach().cpu().clone()) for t in params]
    opt = torch.optim.Adam(ref, lr=1e-3)
    for step in range(2):
        for rp, g in zip(ref, (g3, gl, gt, gb)):
            rp.grad = g.clone()
        opt.step()
        adam.step()
    for t, rp in zip(params, ref):
        check(t, rp.data, 1e-6, "adam + packs: parameters")
    got = [(f.clone(), None if d is None else d.clone()) for f, d, *_ in tab.info]
    tab.run()                                                  # the reference packing of the updated parameters
    torch.cuda.synchronize()
    for (f, d, *_), (gf, gd) in zip(tab.info, got):
        assert torch.equal(f, gf), "F pack written by the optimizer differs from cris_pack_weights"
        assert torch.equal(d, gd), "D pack written by the optimizer differs from cris_pack_weights"


def test_zero_bytes():
    for n in (1, 15, 16, 17, 4099, 1 << 20):
        t = torch.full((n + 32,), 7, dtype=torch.uint8, device=DEV)
        ops.zero_(t[:n])
        assert int(t[:n].max()) == 0 and int(t[n:].min()) == 7, n


def test_syncbn_single_exchange_kernels():
    """cris_bn_sync_pack / _unpack: two 'ranks' (row halves) exchange [S1 | S2] about a shared reference in ONE summation
    (the all-reduce, done here by adding the two packed vectors) and recover the statistics of the concatenated batch."""
    M, C_ = 1000, 40
    y = (rnd(M, C_) * 1.7 + 2.5).to(BF).float()
    ref = (y.mean(0) + 0.3 * rnd(C_, seed=9)).to(DEV)            # a running mean that lags the batch mean
    packed = []
    for half in (y[:400], y[400:]):
        n = half.shape[0]
        st = ops.colstats(bf(half), n, C_, 32, DEV)
        merged, mean_l = torch.zeros(2 * C_, device=DEV), torch.empty(C_, device=DEV)
        ops.bn_finalize(st, n, n, torch.ones(C_, device=DEV), torch.zeros(C_, device=DEV), None, None, 0.1, 1e-5, C_, None, None,
                        mean_l, None, merged=merged)
        ops.bn_sync_pack(merged, mean_l, ref, n, C_)
        packed.append(merged)
    g = packed[0] + packed[1]                                     # the all-reduce
    ops.bn_sync_unpack(g, ref, M, C_)
    outs = [torch.empty(C_, device=DEV) for _ in range(4)]
    ops.bn_finalize(None, 400, M, torch.ones(C_, device=DEV), torch.zeros(C_, device=DEV), None, None, 0.1, 1e-5, C_, *outs,
                    global_stats=g)
    check(outs[2], y.double().mean(0), 1e-6, "global mean")
    check(outs[3], torch.rsqrt(y.double().var(0, unbiased=False) + 1e-5), 1e-5, "global invstd")


@pytest.mark.parametrize("d,tol", [(20.0, 2e-4), (100.0, 5e-3)])
def test_syncbn_single_exchange_far_reference(d, tol):
    """the worst case of the single-exchange form: a FRESH BatchNorm (running mean = 0, the reference the moments are taken
    about) whose input sits d standard deviations away from zero.  M2 = S2 - S1^2/N then cancels (1 + d^2) : 1, i.e. the
    variance loses eps_fp32 * (1 + d^2) of relative accuracy: 2.4e-5 at d = 20, 6e-4 at d = 100 (the bounds asserted on
    1/std are 10x that).  Post-ReLU / post-BatchNorm inputs of this network have |mean|/std of order 1; after a few steps the
    running mean tracks the batch mean and d goes to 0."""
    M, C_ = 4000, 64
    y = (rnd(M, C_) * 0.5 + 0.5 * d).to(BF).float()             # std 0.5, mean d * std
    ref = torch.zeros(C_, device=DEV)
    packed = []
    for half in (y[:2000], y[2000:]):
        n = half.shape[0]
        st = ops.colstats(bf(half), n, C_, 32, DEV)
        merged, mean_l = torch.zeros(2 * C_, device=DEV), torch.empty(C_, device=DEV)
        ops.bn_finalize(st, n, n, torch.ones(C_, device=DEV), torch.zeros(C_, device=DEV), None, None, 0.1, 1e-5, C_, None, None,
                        mean_l, None, merged=merged)
        ops.bn_sync_pack(merged, mean_l, ref, n, C_)
        packed.append(merged)
    g = packed[0] + packed[1]
    ops.bn_sync_unpack(g, ref, M, C_)
    outs = [torch.empty(C_, device=DEV) for _ in range(4)]
    ops.bn_finalize(None, 2000, M, torch.ones(C_, device=DEV), torch.zeros(C_, device=DEV), None, None, 0.1, 1e-5, C_, *outs,
                    global_stats=g)
    check(outs[2], y.double().mean(0), 1e-6, "global mean, reference %g std away" % d)
    check(outs[3], torch.rsqrt(y.double().var(0, unbiased=False) + 1e-5), tol, "global invstd, reference %g std away" % d)


def test_layernorm_backward_wide_instantiation():
    """CRIS_LN_BWD_V=0 (read once per process, hence the subprocess): every width on the V = 4 instantiation of the LayerNorm
    backward (the default runs 1 / 2 channel vectors per lane for C <= 512 / 1024); the same checks as test_layernorm_variants"""
    import subprocess
    import sys
    env = dict(os.environ, CRIS_LN_BWD_V="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", "test_layernorm_variants",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]


def test_conv_wgrad_deferred_grouped_reduction(monkeypatch):
    """CRIS_WGRAD_REDUCE_GROUP: the split reductions of large problems wait in the queue and run as one grouped launch at the
    flush; the result equals the immediate reduction bit for bit (same summation order)"""
    monkeypatch.setattr(ops, "_WGRAD_REDUCE_GROUP", True)
    q = ops.WgradQueue()
    jobs = []
    for case in (dict(B=8, H=52, W=52, C=128, N=256, k=3), dict(B=6, H=52, W=52, C=64, N=72, k=1), dict(B=8, H=52, W=52, C=128, N=256, k=3, tile=256)):
        x, dy, g, ref, C_real = _wgrad_problem(case)
        N, k, C_ = case["N"], case["k"], case["C"]
        dWg = torch.full((N, k * k * C_), float("nan"), device=DEV)
        dbias = torch.full((N,), float("nan"), device=DEV)
        dyb, xb = bf(dy), bf(x)
        ops.conv_wgrad(dyb, xb, g, N, dWg, dbias=dbias, queue=q, tile=case.get("tile", 0))
        jobs.append((case, dWg, dbias, dy, ref, C_real, dyb, xb, g))
    assert len(q.reduces) == 3 and not q.items
    q.flush()
    assert not q.reduces
    for case, dWg, dbias, dy, ref, C_real, dyb, xb, g in jobs:
        _wgrad_check(case, dWg, dbias, dy, ref, C_real, "deferred reduction")
        one, ob = torch.empty_like(dWg), torch.empty_like(dbias)
        monkeypatch.setattr(ops, "_WGRAD_REDUCE_GROUP", False)
        ops.conv_wgrad(dyb, xb, g, case["N"], one, dbias=ob, tile=case.get("tile", 0))
        assert torch.equal(one, dWg) and torch.equal(ob, dbias)


def test_small_fp32_sentence_vector_kernels():
    """csrc/smallf32.hip against torch fp32 (pure fp32 arithmetic: 1e-5): the end-of-text rows through the final LayerNorm, the two
    linear forms (+ accumulate, bias, strided output), the weight-gradient outer sums, BatchNorm1d + ReLU forward / backward over 8 rows"""
    B, L, D, E, C_ = 8, 17, 512, 1024, 640
    tok = torch.randint(1, 400, (B, L))
    for b in range(B):
        tok[b, 3 + b] = 49407
        tok[b, 4 + b:] = 0
    x = rnd(B * L, D, seed=1)
    mean, var = x.mean(1), x.var(1, unbiased=False)
    rstd = torch.rsqrt(var + 1e-5)
    gamma, beta = 1 + 0.1 * rnd(D, seed=2), 0.1 * rnd(D, seed=3)
    d = lambda t: t.to(DEV).contiguous()
    rows, eot = torch.empty(B, D, device=DEV), torch.empty(B, dtype=torch.int32, device=DEV)
    ops.eot_gather_ln_f32(d(tok), d(x), d(mean), d(rstd), d(gamma), d(beta), D, rows, eot)
    idx = tok.argmax(-1)
    assert torch.equal(eot.cpu().long(), idx)
    ref_rows = F.layer_norm(x.view(B, L, D)[torch.arange(B), idx], (D,), gamma, beta, 1e-5)
    check(rows, ref_rows, 1e-5, "eot rows through ln_final")
    dxb = bf(rnd(B * L, D, seed=4))
    drows = rnd(B, D, seed=5)
    want = dxb.float().cpu().clone()
    want.view(B, L, D)[torch.arange(B), idx] += drows
    ops.eot_scatter_add_f32(eot, d(drows), B, L, D, dxb)
    assert torch.equal(dxb.cpu(), want.to(BF))
    # linear, [N][K] weight (+ bias, into a column slice of a wider buffer) and [K][N] parameter, accumulate
    A, W, bias = rnd(B, E, seed=6), rnd(2305, E, seed=7) / 32, rnd(2305, seed=8)
    wide = torch.zeros(B, 2312, device=DEV)
    ops.linear_f32_small(d(A), d(W), wide[:, :2305], bias=d(bias))
    check(wide[:, :2305], A @ W.t() + bias, 1e-5, "linear [N][K] + bias")
    assert float(wide[:, 2305:].abs().max()) == 0.0
    P = rnd(D, E, seed=9) / 22
    out = d(rnd(B, E, seed=10))
    base = out.cpu().clone()
    ops.linear_f32_small(rows, d(P), out, w_is_kn=True, accumulate=True)
    check(out, base + ref_rows @ P, 1e-5, "x @ P accumulated")
    dW, db = torch.empty(2305, E, device=DEV), torch.empty(2305, device=DEV)
    dwb = rnd(B, 2312, seed=11)
    ops.outer_sum_f32_small(d(dwb)[:, :2305], d(A), dW, rowsum=db)
    check(dW, dwb[:, :2305].t() @ A, 1e-5, "outer sum")
    check(db, dwb[:, :2305].sum(0), 1e-5, "bias gradient")
    # BatchNorm1d + ReLU over 8 rows against torch autograd
    y = rnd(B, C_, seed=12).requires_grad_(True)
    g_, b_ = (1 + 0.2 * rnd(C_, seed=13)), 0.3 * rnd(C_, seed=14)
    z = torch.relu(F.batch_norm(y, None, None, g_, b_, True, 0.1, 1e-5))
    dz = rnd(B, C_, seed=15)
    z.backward(dz)
    st = ops.colstats_f32_small(d(y.detach()), DEV)
    outs = [torch.empty(C_, device=DEV) for _ in range(4)]
    ops.bn_finalize(st, B, B, d(g_), d(b_), None, None, 0.1, 1e-5, C_, *outs)
    scale, shift, mu, inv = outs
    z32 = torch.empty(B, C_, device=DEV)
    ops.bn_relu_f32_small(d(y.detach()), scale, shift, z32)
    check(z32, z.detach(), 1e-5, "bn1d + relu")
    sums, dy = torch.zeros(2 * C_, device=DEV), torch.empty(B, C_, device=DEV)
    ops.bn_relu_bwd_f32_small(d(dz), d(y.detach()), scale, shift, mu, inv, sums, B, dy)
    check(dy, y.grad, 2e-5, "bn1d + relu backward")
    gm = torch.where(z.detach() > 0, dz, torch.zeros_like(dz))
    check(sums[:C_], gm.sum(0), 1e-5, "d beta")
