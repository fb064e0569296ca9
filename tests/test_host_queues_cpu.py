"""Host logic of round 4 without a GPU (every library launch replaced by a recorder - no arithmetic happens): how ops.GemmQueue groups
independent GEMM problems into launches, and what cris.pytorch_amd.optim.Adam does when it cannot run the fused update."""
import ctypes

import pytest
import torch

from cris.pytorch_amd import hip, ops
from cris.pytorch_amd.ops import Geom


class _Stream:
    cuda_stream = 0


@pytest.fixture
def recorder(monkeypatch):
    log = []
    monkeypatch.setattr(hip, "call", lambda name, *args: log.append((name, args)))
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _Stream())
    return log


def _problem(q, M, N, K, k=1, bias=False, variant=-1):
    hw = int(round(M ** 0.5)) if k == 3 else M
    g = Geom(1, hw, hw, K, 3, 3, 1, 1) if k == 3 else Geom.linear(M, K)
    A = torch.zeros(g.Bn * g.H * g.W, K, dtype=torch.bfloat16)
    W = torch.zeros(N, g.K, dtype=torch.bfloat16)
    out = torch.zeros(g.M, N, dtype=torch.bfloat16)
    ops.conv_gemm(A, W, g, N, out=out, bias=torch.zeros(N) if bias else None, queue=q, variant=variant)


def test_gemm_queue_groups_by_tile_and_epilogue(recorder):
    q = ops.GemmQueue()
    _problem(q, 5408, 512, 512)                       # lean, 8-wave 128x128 tile (172 tiles)
    _problem(q, 5408, 512, 1024)                      # the same key
    _problem(q, 5408, 512, 512, bias=True)            # lean + bias: another epilogue instantiation -> its own launch
    _problem(q, 1352, 512, 512)                       # too few tiles for the 8-wave tile: 64x64
    _problem(q, 8, 1024, 1024)                        # M = 8: a skinny kernel, never grouped
    assert len(q) == 5 and not recorder
    q.flush()
    names = [n for n, _ in recorder]
    assert names.count("cris_conv_gemm_group_launch") == 1 and names.count("cris_conv_gemm_variant") == 3
    grp = next(a for n, a in recorder if n == "cris_conv_gemm_group_launch")[0]._obj
    assert grp.n == 2 and [grp.prob[i].K for i in range(2)] == [1024, 512]          # longest reduction first
    variants = ops.gemm_variants()
    assert variants[next(a for n, a in recorder if n == "cris_conv_gemm_group_launch")[1]] == "8w128x128"
    assert len(q) == 0


def test_gemm_queue_chunks_and_forced_variant(recorder):
    q = ops.GemmQueue()
    for i in range(hip.GEMM_GROUP_MAX + 3):
        _problem(q, 1352, 256, 256 + 64 * (i % 4), variant="64x64")
    q.flush()
    groups = [a[0]._obj.n for n, a in recorder if n == "cris_conv_gemm_group_launch"]
    assert groups == [hip.GEMM_GROUP_MAX, 3]
    del recorder[:]
    q2 = ops.GemmQueue()
    q2.enabled = False                                # CRIS_GEMM_GROUPS=0: every problem alone
    _problem(q2, 5408, 512, 512)
    _problem(q2, 5408, 512, 1024)
    q2.flush()
    assert [n for n, _ in recorder] == ["cris_conv_gemm_variant"] * 2


def test_bn_backward_partials_are_refused_for_a_queued_launch(recorder):
    q = ops.GemmQueue()
    g = Geom.linear(1352, 256)
    A, W, out = torch.zeros(1352, 256, dtype=torch.bfloat16), torch.zeros(256, 256, dtype=torch.bfloat16), torch.zeros(1352, 256, dtype=torch.bfloat16)
    y = torch.zeros(1352, 256, dtype=torch.bfloat16)
    v = torch.zeros(256)
    with pytest.raises(AssertionError):
        ops.conv_gemm(A, W, g, 256, out=out, queue=q, bnr=dict(y=y, ldy=256, coff=0, mean=v, invstd=v, scale=v, shift=v))
    parts = ops.conv_gemm(A, W, g, 256, out=out, bnr=dict(y=y, ldy=256, coff=0, mean=v, invstd=v, scale=v, shift=v))
    assert isinstance(parts, ops.BnrParts) and parts.t.shape == (parts.nparts, 512)
    p = recorder[-1][1][0]._obj
    assert p.bnr_y == y.data_ptr() and p.stat_ld == 512 and p.colsq == p.colsum + 4 * 256
    # a list of more than BNR_MAX_PARTS row blocks: plain launch, the caller reduces in a launch of its own
    big = Geom.linear(32 * (ops.BNR_MAX_PARTS + 1), 64)
    Ab, Wb, ob = torch.zeros(big.M, 64, dtype=torch.bfloat16), torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(big.M, 64, dtype=torch.bfloat16)
    assert ops.conv_gemm(Ab, Wb, big, 64, out=ob, variant="64x64", bnr=dict(y=ob, ldy=64, coff=0, mean=v, invstd=v, scale=v, shift=v)) is None
    assert not recorder[-1][1][0]._obj.bnr_y


def test_optional_adam_is_torch_adam_without_an_engine_backed_module():
    """parameters that belong to no CRIS module: cris.pytorch_amd.optim.Adam behaves as torch.optim.Adam, GradScaler sees an ordinary
    optimizer (`_step_supports_amp_scaling` False), the results are torch's bit for bit"""
    from cris.pytorch_amd import optim
    torch.manual_seed(0)
    w0 = torch.randn(7, 5)
    res = []
    for cls in (optim.Adam, torch.optim.Adam):
        w = torch.nn.Parameter(w0.clone())
        opt = cls([w], lr=1e-2)
        assert not getattr(opt, "_step_supports_amp_scaling", False)
        for i in range(3):
            opt.zero_grad()
            ((w * w).sum() + w.sum() * i).backward()
            opt.step()
        res.append(w.detach().clone())
    assert torch.equal(res[0], res[1])
