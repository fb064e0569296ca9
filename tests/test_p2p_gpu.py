"""The peer-mailbox SyncBN exchange
(csrc/p2p.hip, dist.PeerMailboxes) with two ranks sharing the one GPU of the test box - the mailboxes travel through HIP
IPC exactly as between two GPUs.  (1) the primitive: sums over ranks, slot / generation reuse, device generation counter;
(2) the whole trainer with CRIS_SYNCBN_P2P=1 equals the run that exchanges through torch.distributed."""
import dataclasses
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

gpu = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _prim_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cris.pytorch_amd.dist import TorchDistComm
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        comm = TorchDistComm(dev)
        gen = torch.zeros(1, dtype=torch.int32, device=dev)
        why = comm.enable_p2p(slots=4, max_floats=1000, gen_dev=gen)        # allocation, IPC mapping, self-test of both protocols
        assert why is None, why
        box = comm.p2p
        ok = True
        for step in range(6):                                   # generations 0..5: both parities, every slot reused
            gen.fill_(step)
            for slot, n in enumerate((1, 64, 999, 1000)):
                for proto in (box.allreduce_sum, box.ll_allreduce_sum):      # the flag-protocol kernel, then the LL words
                    g = torch.Generator().manual_seed(100 * step + slot)
                    full = torch.randn(world, n, generator=g)
                    t = full[rank].clone().to(dev)
                    proto(t, slot, gen_dev=gen if step % 2 else None, gen_host=step)
                    want = full[0].clone()
                    for r in range(1, world):
                        want = want + full[r]                   # rank order, like the kernels
                    ok = ok and torch.equal(t.cpu(), want)
        torch.cuda.synchronize()
        ok = ok and int(box.err.item()) == 0
        dist.barrier()
        box.close()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@gpu
def test_peer_mailbox_allreduce_two_ranks_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_prim_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def _train_worker(rank, world, port, p2p, launch, q):
    """p2p: False = exchange through torch.distributed; "kernel" = mailboxes, the exchange a kernel of its own between the
    BatchNorm launches (CRIS_SYNCBN_FUSED=0); True = mailboxes, the exchange inside the BatchNorm launches (the default)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CRIS_SYNCBN_P2P"] = "1" if p2p else "0"
    os.environ["CRIS_SYNCBN_FUSED"] = "0" if p2p == "kernel" else "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cris.pytorch_amd import arch, synth
        from cris.pytorch_amd.dist import TorchDistComm
        from cris.pytorch_amd.trainer import NativeTrainer
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        clip, head = arch.specs_by_name("tiny")
        head = dataclasses.replace(head, dropout=0.0)      # mask indices are rank-local: compare without dropout
        tr = NativeTrainer(clip, head, arch.synthetic_state_dict(clip, head, 0), dev, comm=TorchDistComm(dev), sync_bn=True,
                           launch=launch)
        assert (tr.comm.p2p is not None) == bool(p2p)
        losses = []
        for step in range(4):
            img, word, mask = (t.to(dev) for t in synth.make_batch(4, 64, 9, rank, step))
            loss, _ = tr.train_step(img, word, mask)
            losses.append(float(loss))
        torch.cuda.synchronize()
        err = int(tr.comm.p2p.err.item()) if p2p else 0
        import hashlib
        h = hashlib.sha256()
        for k in sorted(tr.engine.P):
            h.update(tr.engine.P[k].detach().cpu().numpy().tobytes())
        for k in sorted(tr.engine.Bf):
            h.update(tr.engine.Bf[k].detach().cpu().numpy().tobytes())
        q.put((rank, losses, err, h.hexdigest()))
    finally:
        dist.destroy_process_group()


# (two processes time-slice the one GPU of the test box - every exchange costs a scheduling quantum, 50 - 80 s per variant: the
# command-list variant runs with the long parity runs, `-m "gpu or gpu_long"`)
@pytest.mark.parametrize("launch", [pytest.param("eager", marks=gpu), pytest.param("cmdlist", marks=pytest.mark.gpu_long)])
def test_trainer_with_peer_mailboxes_equals_torch_distributed_exchange(launch):
    """two ranks, four optimizer steps: the SyncBN exchange through torch.distributed, through the mailbox kernel, and INSIDE the
    BatchNorm launches (the default) - bit-identical losses, and bit-identical parameters and running statistics at the end (two
    ranks: a + b in either order; the pack / unpack arithmetic is the same code in all three forms)"""
    ctx = mp.get_context("spawn")
    out = {}
    modes = (False, "kernel", True) if launch == "eager" else (False, True)        # (the stand-alone exchange kernel: eager only)
    for p2p in modes:
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_train_worker, args=(r, 2, port, p2p, launch, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=600) for _ in procs)
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        out[p2p] = res
    for mode in modes[1:]:
        for (_, la, _, ha), (_, lb, err, hb) in zip(out[False], out[mode]):
            assert err == 0
            assert la == lb, (mode, la, lb)
            assert ha == hb, "parameters / running statistics differ from the torch.distributed run (%r)" % (mode,)
    assert out[True][0][3] == out[True][1][3]                  # and both ranks hold the same model
