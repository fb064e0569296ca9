"""The reference's recipe, end to end, on the drop-in module (train.py:96-111 + engine/engine.py:37-73, closing SURVEY.md 8(a)
rows a14 / a16): `build_segmenter(args)` -> `nn.SyncBatchNorm.convert_sync_batchnorm` -> `DistributedDataParallel(model.cuda(),
device_ids=[gpu], find_unused_parameters=True)` -> Adam over the two parameter groups -> MultiStepLR -> GradScaler, driven by the
loop body of engine.train (tests/ref_loop.py, pinned to the reference's own function by tests/test_ref_loop_cpu.py): ambient
fp16 autocast, scaled backward, scaler.step / update, trainMetricGPU, three scalar all-reduces.

Two ranks share the one GPU of the test box (gloo carries DDP's buckets, SyncBN's statistics and the scalars; RCCL needs one
GPU per rank).  Checked: (1) the 2-rank run equals the 1-rank run of the same recipe on the concatenated batch - the property
DDP + SyncBN guarantee; (2) the first loss equals the fp32 CPU oracle's on the concatenated batch; (3) ranks stay in
lock-step (identical parameters and running statistics)."""
import dataclasses
import os
import socket
from types import SimpleNamespace as NS

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

STEPS = 3
PROBES = ("backbone.visual.layer1.0.conv2.weight", "neck.f2_cat.1.weight", "decoder.layers.0.ffn.0.weight", "proj.txt.weight")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cfg():
    from test_module_surface import TINY
    return NS(**dict(TINY, dropout=0.0))                     # dropout masks are indexed per rank: compare without


def _shards(world_total, rank, world, step):
    """rank's quarter/half of the step's global batch of 8: the 1-rank run sees the concatenation of the 2 ranks' shards"""
    from cris.pytorch_amd import synth
    parts = [synth.make_batch(4, 64, 9, r, step) for r in range(world_total)]
    if world == 1:
        img, word, mask = (torch.cat([p[i] for p in parts]) for i in range(3))
    else:
        img, word, mask = parts[rank]
    return img, word, mask[:, 0]                              # the loader yields [B,H,W] masks; the loop adds the channel dim


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import ref_loop
        from torch import nn
        from cris.pytorch_amd import arch
        from cris.pytorch_amd.model import build_segmenter
        torch.cuda.set_device(0)
        args = _cfg()
        model, param_list = build_segmenter(args)                                  # train.py:96
        clip, head = arch.specs_by_name("tiny")
        model.load_state_dict(arch.synthetic_state_dict(clip, dataclasses.replace(head, dropout=0.0), 0))
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)                     # train.py:97-98
        model = nn.parallel.DistributedDataParallel(model.cuda(), device_ids=[0], find_unused_parameters=True)    # :100-102
        optimizer = torch.optim.Adam(param_list, lr=args.base_lr, weight_decay=0.0)      # train.py:105-107
        scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[35], gamma=0.1)      # noqa: F841 (:108-110)
        scaler = torch.amp.GradScaler("cuda")                                      # train.py:111
        batches = [_shards(2, rank, world, s) for s in range(STEPS)]
        res = ref_loop.train_steps(batches, model, optimizer, scaler)
        torch.cuda.synchronize()
        sd = model.module.state_dict()
        probe = {k: sd[k].double().sum().item() for k in PROBES}
        rm = sd["backbone.visual.bn1.running_mean"].double().sum().item()
        q.put((rank, res, probe, rm, float(scaler.get_scale())))
    finally:
        dist.destroy_process_group()


def _run(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


def test_reference_recipe_two_ranks_equal_one_rank_and_the_oracle():
    two = _run(2)
    one = _run(1)
    (_, r0, p0, rm0, s0), (_, r1, p1, rm1, s1) = two
    (_, ref, pref, rmref, sref) = one[0]
    print("2 ranks:", r0, "1 rank:", ref)
    assert r0 == r1                                           # the three all-reduces make the reported figures identical
    for step in range(STEPS):
        assert abs(r0[step][0] - ref[step][0]) < 1e-2, (step, r0[step], ref[step])        # loss
        assert abs(r0[step][1] - ref[step][1]) < 5.0, (step, r0[step], ref[step])         # IoU in % (thresholded: coarse)
    for k in PROBES:                                          # ranks in lock-step; the 2-rank run tracks the 1-rank run
        assert abs(p0[k] - p1[k]) <= 1e-6 * max(1.0, abs(p0[k])), (k, p0[k], p1[k])
        assert abs(p0[k] - pref[k]) <= 1e-2 * max(1.0, abs(pref[k])), (k, p0[k], pref[k])
    assert abs(rm0 - rm1) < 1e-6 and abs(rm0 - rmref) < 1e-3 * max(1.0, abs(rmref))          # SyncBN: global running statistics
    assert s0 == s1 == sref                                   # GradScaler saw finite gradients on every rank: same scale
    # first loss against the fp32 CPU oracle on the concatenated batch (before any update)
    from cris.pytorch_amd import arch
    from oracle import cris_oracle as O
    clip, head = arch.specs_by_name("tiny")
    head = dataclasses.replace(head, dropout=0.0)
    sd = arch.synthetic_state_dict(clip, head, 0)
    img, word, mask = _shards(2, 0, 1, 0)
    with torch.no_grad():
        _, _, oloss = O.cris_forward(sd, clip, head, img, word, mask.unsqueeze(1), training=True, drop_seed=None)
    assert abs(ref[0][0] - float(oloss)) < 1e-2 and abs(r0[0][0] - float(oloss)) < 1e-2, (ref[0], r0[0], float(oloss))


# ---------------------------------------------------------------------------------------------------------------------------
# SURVEY.md 8e option B: under DistributedDataParallel the module names its parameters in `_ddp_params_and_buffers_to_ignore`
# and exchanges the gradient arena itself (model/segmenter.py).  Pinned against the form of rounds 2-5, in which DDP manages
# every parameter (CRIS_DDP_SELF_EXCHANGE=0): the averaged gradients must be the SAME BITS - the module divides by the world
# size before the sum like DDP's reducer, a division by 2 is exact, and a two-term sum has one order.
# ---------------------------------------------------------------------------------------------------------------------------
def _grad_worker(rank, world, port, q, self_exchange, optimizer_name, grad_exchange="rccl"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CRIS_DDP_SELF_EXCHANGE"] = "1" if self_exchange else "0"
    os.environ["CRIS_GRAD_EXCHANGE"] = grad_exchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import hashlib
        from torch import nn
        from cris.pytorch_amd import arch
        from cris.pytorch_amd.model import build_segmenter
        torch.cuda.set_device(0)
        args = _cfg()
        model, param_list = build_segmenter(args)
        clip, head = arch.specs_by_name("tiny")
        model.load_state_dict(arch.synthetic_state_dict(clip, dataclasses.replace(head, dropout=0.0), 0))
        if rank == 1:
            # DDP starts every rank from rank 0's parameters (the head is randomly initialised per process in the reference):
            # whoever manages the parameters has to repair this
            with torch.no_grad():
                model.proj.txt.weight.add_(0.5)
                model.backbone.visual.layer1[0].bn1.running_mean.add_(0.25)
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
        model = nn.parallel.DistributedDataParallel(model.cuda(), device_ids=[0], find_unused_parameters=True)
        inner = model.module
        managed = len(list(model._module_parameters)) if hasattr(model, "_module_parameters") else -1
        if optimizer_name == "cris":
            from cris.pytorch_amd import optim as cris_optim
            optimizer = cris_optim.Adam(param_list, lr=args.base_lr, weight_decay=0.0)
        else:
            optimizer = torch.optim.Adam(param_list, lr=args.base_lr, weight_decay=0.0)
        scaler = torch.amp.GradScaler("cuda")
        model.train()
        out = {"managed_by_ddp": managed, "self_exchange": None, "digests": [], "losses": []}

        def digest():
            h = hashlib.sha256()
            tot = 0.0
            for n, p in inner.named_parameters():
                if p.grad is None:
                    assert n == "backbone.logit_scale", n
                    continue
                g = p.grad.detach().float().contiguous().cpu()
                h.update(g.numpy().tobytes())
                tot += float(g.double().abs().sum())
            return h.hexdigest(), tot

        for step in range(STEPS):
            image, text, target = (t.cuda() for t in _shards(2, rank, world, step))
            with torch.autocast("cuda"):
                pred, tgt, loss = model(image, text, target.unsqueeze(1))
            optimizer.zero_grad()
            scaler.scale(loss).backward()
            out["digests"].append(digest())
            scaler.step(optimizer)
            scaler.update()
            out["losses"].append(float(loss))
        # gradient accumulation through the wrapper's no_sync(): first micro-batch local, second exchanged
        image, text, target = (t.cuda() for t in _shards(2, rank, world, STEPS))
        optimizer.zero_grad()
        with model.no_sync():
            with torch.autocast("cuda"):
                _, _, loss = model(image, text, target.unsqueeze(1))
            scaler.scale(loss).backward()
        image, text, target = (t.cuda() for t in _shards(2, rank, world, STEPS + 1))
        with torch.autocast("cuda"):
            _, _, loss = model(image, text, target.unsqueeze(1))
        scaler.scale(loss).backward()
        # (a numpy array: pickled by value - a torch tensor travels through the queue as a file descriptor that the parent can
        # only pick up while this process is alive)
        acc = torch.cat([p.grad.detach().float().flatten() for n, p in inner.named_parameters() if p.grad is not None]).cpu().numpy()
        out["self_exchange"] = bool(getattr(inner, "_self_exchange", False))
        out["syncbn_exchange"] = getattr(inner, "syncbn_exchange", None)
        out["grad_exchange"] = getattr(inner, "grad_exchange", None)
        torch.cuda.synchronize()
        sd = inner.state_dict()
        out["probe"] = {k: sd[k].double().sum().item() for k in PROBES}
        out["rm"] = sd["backbone.visual.layer1.0.bn1.running_mean"].double().sum().item()
        q.put((rank, out, acc))
    finally:
        dist.destroy_process_group()


def _run_grad(self_exchange, optimizer_name="torch", grad_exchange="rccl"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q, self_exchange, optimizer_name, grad_exchange)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=900) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("optimizer_name", ["torch", "cris"])
def test_own_gradient_exchange_under_ddp_equals_ddp_managed_gradients_bit_for_bit(optimizer_name):
    own = _run_grad(True, optimizer_name)
    ddp = _run_grad(False, optimizer_name)
    (_, o0, a0), (_, o1, a1) = own
    (_, d0, b0), (_, d1, b1) = ddp
    assert o0["self_exchange"] and o1["self_exchange"] and not d0["self_exchange"]
    # (round 6: the drop-in module's SyncBN statistics travel through the peer mailboxes, inside the BatchNorm launches)
    assert "p2p mailboxes" in o0["syncbn_exchange"] and "p2p mailboxes" in d0["syncbn_exchange"], (o0["syncbn_exchange"], d0["syncbn_exchange"])
    assert o0["managed_by_ddp"] == 1 and d0["managed_by_ddp"] > 100, (o0["managed_by_ddp"], d0["managed_by_ddp"])
    print("own", o0["losses"], "ddp", d0["losses"])
    # every rank holds the same averaged gradient, and it is the one DDP's reducer produces - the same bits, every step (the
    # later steps also cover the optimizer having seen the same gradients and rank 1's repaired parameters)
    assert o0["digests"] == o1["digests"]
    assert d0["digests"] == d1["digests"]
    assert o0["digests"] == d0["digests"], (o0["digests"], d0["digests"])
    assert o0["losses"] == d0["losses"] and o1["losses"] == d1["losses"]
    assert o0["probe"] == o1["probe"] == d0["probe"] and o0["rm"] == o1["rm"] == d0["rm"]
    # no_sync() accumulation: old (local) + new (exchanged) parts are averaged separately here, together by DDP - equal up to
    # the rounding of one addition
    import numpy as np
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1)
    rel = float(np.linalg.norm(a0.astype(np.float64) - b0) / np.linalg.norm(b0.astype(np.float64)))
    assert rel < 1e-6, rel


def test_own_gradient_exchange_over_the_mapped_arenas_equals_ddp_managed_gradients():
    """the same pin for the opt-in direct exchange (CRIS_GRAD_EXCHANGE=p2p: reduce-scatter + all-gather over the IPC-mapped gradient
    arenas, csrc/p2p.hip) under the DDP wrapper: with two ranks its rank-order sum (g0 / 2 + g1 / 2) has the bits of DDP's"""
    own = _run_grad(True, "cris", "p2p")
    ddp = _run_grad(False, "cris")
    (_, o0, a0), (_, o1, a1) = own
    (_, d0, b0), (_, d1, b1) = ddp
    assert o0["grad_exchange"] == "p2p" and o1["grad_exchange"] == "p2p", (o0["grad_exchange"], o1["grad_exchange"])
    assert o0["self_exchange"] and not d0["self_exchange"]
    assert o0["digests"] == o1["digests"] == d0["digests"] == d1["digests"]
    assert o0["losses"] == d0["losses"] and o0["probe"] == d0["probe"] and o0["rm"] == d0["rm"]
