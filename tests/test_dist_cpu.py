"""World-size-2 `gloo` tests (CPU) of the data-parallel host logic in cris/pytorch_amd/dist.py: the SyncBN statistic
merge (sum / M2 exchange == statistics of the concatenated batch, reference train.py:97-98 semantics) and the staged
asynchronous gradient all-reduce (== DDP's averaged gradients, reference train.py:100-102)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cris.pytorch_amd.dist import TorchDistComm, merge_batchnorm_partials
        from cris.pytorch_amd import synth
        comm = TorchDistComm(None)
        assert comm.world == world and comm.rank == rank
        # --- SyncBN forward exchange: each rank holds its own rows of a [world*n, C] activation
        g = torch.Generator().manual_seed(7)
        full = torch.randn(world * 50, 16, generator=g, dtype=torch.float64) * 3 + 5       # large mean: cancellation-prone
        mine = full[rank * 50:(rank + 1) * 50]
        s = mine.sum(0)
        m2 = ((mine - mine.mean(0)) ** 2).sum(0)
        mean_g, var_g = merge_batchnorm_partials(s, m2, 50.0, comm, ref=torch.full((16,), 4.5, dtype=torch.float64))
        ok_bn = torch.allclose(mean_g, full.mean(0), atol=1e-12) and torch.allclose(var_g, full.var(0, unbiased=False), atol=1e-12)
        # --- staged gradient exchange over a flat arena, issued back to front like backward does
        arena = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        stages = [(0, 300), (300, 640), (640, 1000)]
        for lo, hi in reversed(stages):
            comm.allreduce_async(arena[lo:hi])
        comm.wait_all()
        expect = torch.arange(1000, dtype=torch.float32) * sum(r + 1 for r in range(world))
        ok_ar = torch.equal(arena, expect)
        # --- per-rank synthetic shards differ (sample sharding, no overlap in seeds)
        a = synth.make_batch(2, 32, 9, rank, 0)[0]
        t = [torch.zeros_like(a) for _ in range(world)]
        dist.all_gather(t, a)
        ok_shard = not torch.equal(t[0], t[1])
        q.put((rank, ok_bn, ok_ar, ok_shard))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_syncbn_merge_and_grad_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok_bn, ok_ar, ok_shard in res:
        assert ok_bn, "SyncBN merge != concatenated-batch statistics on rank %d" % rank
        assert ok_ar, "staged all-reduce wrong on rank %d" % rank
        assert ok_shard


def test_single_exchange_cancellation_bound_fp32():
    """the SyncBN single-exchange arithmetic in fp32 with the reference d standard deviations from the batch mean (a fresh
    BatchNorm: running mean 0): relative variance error ~ eps * (1 + d^2)  (see the GPU test of cris_bn_sync_unpack)."""
    from cris.pytorch_amd.dist import merge_batchnorm_partials

    class One:
        world = 2
        def __init__(self):
            self.buf = None
        def allreduce_sum(self, t):
            t.mul_(2.0)                                    # two identical ranks

    g = torch.Generator().manual_seed(0)
    x = torch.randn(2000, 32, generator=g, dtype=torch.float64) * 0.5
    for d, tol in ((20.0, 2e-4), (100.0, 5e-3)):
        y = x + 0.5 * d
        s = y.sum(0).float()
        m2 = ((y - y.mean(0)) ** 2).sum(0).float()
        mean_g, var_g = merge_batchnorm_partials(s, m2, 2000.0, One(), ref=torch.zeros(32))
        ref_var = y.var(0, unbiased=False)
        assert float(((var_g.double() - ref_var) / ref_var).abs().max()) < tol, d
        assert float((mean_g.double() - y.mean(0)).abs().max()) < 1e-4
