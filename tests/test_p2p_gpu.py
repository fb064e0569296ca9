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

pytestmark = [pytest.mark.gpu]


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
        from cris.pytorch_amd.dist import PeerMailboxes
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        box = PeerMailboxes(rank, world, dev, slots=4, max_floats=1000)
        ok = True
        gen = torch.zeros(1, dtype=torch.int32, device=dev)
        for step in range(6):                                   # generations 0..5: both parities, every slot reused
            gen.fill_(step)
            for slot, n in enumerate((1, 64, 999, 1000)):
                g = torch.Generator().manual_seed(100 * step + slot)
                full = torch.randn(world, n, generator=g)
                t = full[rank].clone().to(dev)
                box.allreduce_sum(t, slot, gen_dev=gen if step % 2 else None, gen_host=step)
                want = full[0].clone()
                for r in range(1, world):
                    want = want + full[r]                       # rank order, like the kernel
                ok = ok and torch.equal(t.cpu(), want)
        torch.cuda.synchronize()
        ok = ok and int(box.err.item()) == 0
        dist.barrier()
        box.close()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


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
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CRIS_SYNCBN_P2P"] = "1" if p2p else "0"
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
        assert (tr.comm.p2p is not None) == p2p
        losses = []
        for step in range(4):
            img, word, mask = (t.to(dev) for t in synth.make_batch(4, 64, 9, rank, step))
            loss, _ = tr.train_step(img, word, mask)
            losses.append(float(loss))
        torch.cuda.synchronize()
        err = int(tr.comm.p2p.err.item()) if p2p else 0
        q.put((rank, losses, err))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("launch", ["eager", "cmdlist"])
def test_trainer_with_peer_mailboxes_equals_torch_distributed_exchange(launch):
    ctx = mp.get_context("spawn")
    out = {}
    for p2p in (False, True):
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
    for (_, la, _), (_, lb, err) in zip(out[False], out[True]):
        assert err == 0
        for a, b in zip(la, lb):
            assert abs(a - b) < 1e-2, (la, lb)          # same arithmetic up to the summation order of the exchange
