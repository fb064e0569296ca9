"""Two ranks on ONE GPU (gloo carries the collectives; NCCL/RCCL needs one GPU per rank, the gpurun box has one): the
data-parallel path for real - SyncBN statistics exchange in forward and backward, staged gradient all-reduce, 1/world in
Adam - against the property the reference's DDP + SyncBN recipe guarantees (train.py:97-102): an N-rank run equals the
1-rank run on the concatenated batch."""
import dataclasses
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard(rank, step):
    from cris.pytorch_amd import synth
    return synth.make_batch(4, 64, 9, rank, step)


def _worker(rank, world, port, launch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cris.pytorch_amd import arch
        from cris.pytorch_amd.dist import TorchDistComm
        from cris.pytorch_amd.trainer import NativeTrainer
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        clip, head = arch.specs_by_name("tiny")
        head = dataclasses.replace(head, dropout=0.0)      # mask indices are rank-local: compare without dropout
        sd = arch.synthetic_state_dict(clip, head, 0)
        tr = NativeTrainer(clip, head, sd, dev, comm=TorchDistComm(dev), sync_bn=True, launch=launch)
        losses = []
        for step in range(STEPS):
            img, word, mask = (t.to(dev) for t in _shard(rank, step))
            loss, _ = tr.train_step(img, word, mask)
            losses.append(float(loss))
        torch.cuda.synchronize()
        probe = {k: tr.engine.P[k].double().sum().item() for k in ("backbone.visual.layer1.0.conv2.weight", "neck.f2_cat.1.weight",
                                                                     "decoder.layers.0.ffn.0.weight", "proj.txt.weight")}
        rm = tr.engine.Bf["backbone.visual.bn1.running_mean"].double().sum().item()
        q.put((rank, losses, probe, rm, tr.launch))
    finally:
        dist.destroy_process_group()


def _single():
    from cris.pytorch_amd import arch
    from cris.pytorch_amd.trainer import NativeTrainer
    dev = torch.device("cuda:0")
    clip, head = arch.specs_by_name("tiny")
    head = dataclasses.replace(head, dropout=0.0)      # mask indices are rank-local: compare without dropout
    sd = arch.synthetic_state_dict(clip, head, 0)
    tr = NativeTrainer(clip, head, sd, dev, launch="eager")
    losses = []
    for step in range(STEPS):
        parts = [_shard(r, step) for r in range(2)]
        img, word, mask = (torch.cat([p[i] for p in parts]).to(dev) for i in range(3))
        loss, _ = tr.train_step(img, word, mask)
        losses.append(float(loss))
    torch.cuda.synchronize()
    probe = {k: tr.engine.P[k].double().sum().item() for k in ("backbone.visual.layer1.0.conv2.weight", "neck.f2_cat.1.weight",
                                                                 "decoder.layers.0.ffn.0.weight", "proj.txt.weight")}
    rm = tr.engine.Bf["backbone.visual.bn1.running_mean"].double().sum().item()
    return losses, probe, rm


@pytest.mark.parametrize("launch", ["eager", "cmdlist"])
def test_two_ranks_equal_one_rank_on_the_concatenated_batch(launch):
    # dropout 0 for this comparison (mask indices are rank-local); everything else as in training
    try:
        ref_losses, ref_probe, ref_rm = _single()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, launch, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=600) for _ in procs)
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
    finally:
        pass
    (_, l0, p0, rm0, m0), (_, l1, p1, rm1, m1) = res
    assert m0 == launch and m1 == launch
    print("rank losses", l0, l1, "single", ref_losses)
    for step in range(STEPS):
        both = 0.5 * (l0[step] + l1[step])                       # mean BCE over 8 samples = mean of the two 4-sample means
        assert abs(both - ref_losses[step]) < 1e-2, (step, both, ref_losses[step])      # observed 1e-3 ... 3e-3
    for k in ref_probe:                                          # parameters stay in lock-step across ranks and track the 1-rank run
        assert abs(p0[k] - p1[k]) <= 1e-6 * max(1.0, abs(p0[k])), (k, p0[k], p1[k])
        # (a sum over ~1e5 parameters that each moved by +-lr per step: Adam's sign noise allows ~1e-2 of the sum)
        assert abs(p0[k] - ref_probe[k]) <= 1e-2 * max(1.0, abs(ref_probe[k])), (k, p0[k], ref_probe[k])
    assert abs(rm0 - rm1) < 1e-6 and abs(rm0 - ref_rm) < 1e-3 * max(1.0, abs(ref_rm))     # SyncBN: global running stats


def test_rccl_collectives_inside_the_captured_graph():
    """The multi-rank code paths (142 SyncBN all-reduces per step on the compute stream, 8 staged gradient all-reduces on their
    own communicator and stream) over a ONE-rank RCCL group - the only RCCL this one-GPU box can run: the step captured into
    a HIP graph (the default launch mode on N ranks) and the command-list replay give the same losses as plain eager launches,
    and the capture reports no error (tools/dist1_check.py)."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    # ("graph+p2p": the same capture with the gradients through the opt-in direct exchange over the mapped arenas - 24 more kernel
    # nodes instead of 8 RCCL collectives; at a world of one it must not change a bit)
    for i, mode in enumerate(("graph", "cmdlist", "eager", "graph+p2p")):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), CRIS_GRAD_EXCHANGE="p2p" if mode.endswith("+p2p") else "rccl")
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "dist1_check.py"), mode.split("+")[0], "4"], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=600)
        out = r.stdout.decode()
        line = [x for x in out.splitlines() if x.startswith("mode ")]
        # (the FIRST 2 kB: an abort's cause is printed before the stack dumps that follow it)
        assert r.returncode == 0 and line, "rc %d\n%s\n...\n%s" % (r.returncode, out[:2000], out[-1000:])
        m = re.match(r"mode (\w+) -> launch (\w+) graph_error (.*?) syncbn (.*?) \| .* losses (\[.*?\]) \.\.\. ([0-9.]+)", line[0])
        assert m, line[0]
        assert m.group(2) == mode.split("+")[0] and m.group(3) == "None", line[0]
        assert ("gradients: p2p" in line[0]) == mode.endswith("+p2p"), line[0]
        # (with a world of one the mailboxes are in use as well: the BatchNorm kernels exchange with themselves inside their launches)
        assert "inside the BatchNorm launches" in m.group(4), line[0]
        got[mode] = (m.group(5), m.group(6))
    assert got["graph"] == got["cmdlist"] == got["eager"] == got["graph+p2p"], got


def test_rccl_drop_in_module_under_ddp_with_the_multi_rank_paths_forced():
    """The drop-in module under the reference's recipe (one-rank RCCL group + SyncBatchNorm + DistributedDataParallel, bench.py
    --path module --ddp-one-rank) with CRIS_FORCE_DIST=1: what one GPU can run of the N-rank module path - the module's own
    gradient exchange (eight all-reduces on their own communicator and stream) captured INSIDE the backward HIP graph while
    DistributedDataParallel's process group and its c10d watchdog are alive, SyncBatchNorm's statistics through the peer
    mailboxes inside the BatchNorm launches of both captured graphs.  Against the same command without the switch (no exchange
    at a world of one): no capture error, the same losses up to the rounding of the single-exchange statistics."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    recs = {}
    for forced in (True, False):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "CRIS_FORCE_DIST")}
        if forced:
            env["CRIS_FORCE_DIST"] = "1"
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--path", "module", "--ddp-one-rank", "--optimizer", "cris",
                            "--steps", "4", "--warmup", "3"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert r.returncode == 0 and lines, "rc %d\n%s\n...\n%s" % (r.returncode, r.stderr[:2000], r.stderr[-1500:])
        recs[forced] = json.loads(lines[0])["config"]
    f, p = recs[True], recs[False]
    print("forced", f, "plain", p)
    assert f["graph_error"] is None and p["graph_error"] is None
    assert f["own_gradient_exchange"] and f["gradient_exchange_in_this_run"] and f["ddp_managed_parameters"] == 1
    assert p["own_gradient_exchange"] and not p["gradient_exchange_in_this_run"]
    assert "inside the BatchNorm launches" in f["syncbn_exchange"], f["syncbn_exchange"]
    # (the same seeded random head in both runs; its untrained logits are huge - sum |logit| ~ 2e5 - and turn the rounding of the
    # single-exchange statistics, moments about the running mean instead of the two-pass merge, into ~1e-2 of loss at step 0:
    # tools/forced_sync_debug.py, call r06d: 1.8425 plain, 1.8502 / 1.8596 forced; after seven chaotic steps only the regime is compared)
    assert abs(f["first_loss"] - p["first_loss"]) < 5e-2 and abs(f["final_loss"] - p["final_loss"]) < 0.5, (f, p)
    import math
    assert math.isfinite(f["final_loss"]) and f["grad_scale"] == p["grad_scale"]
