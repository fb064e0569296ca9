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
