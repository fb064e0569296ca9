"""The library-owned RCCL communicators (csrc/comm.hip, dist.RcclComm) with ONE rank on one GPU - the gpurun box has a single
GPU and RCCL wants one GPU per rank: exchanges are identities, what is checked is that RCCL resolves and initialises from
inside libcris_hip.so, that both communicators / the side stream / the events work, that the multi-rank code paths of the
trainer (SyncBN exchanges, staged gradient all-reduce, broadcast at construction) run on it and that the whole step including
RCCL's kernels is captured into ONE HIP graph.  Prints one JSON line.   python tools/comm1_check.py [spec] [steps]"""
import json
import os
import sys
import time

os.environ["CRIS_FORCE_DIST"] = "1"
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import arch, synth                       # noqa: E402
from cris.pytorch_amd.dist import RcclComm                      # noqa: E402
from cris.pytorch_amd.trainer import NativeTrainer              # noqa: E402

spec, steps = (sys.argv[1] if len(sys.argv) > 1 else "tiny"), int(sys.argv[2]) if len(sys.argv) > 2 else 6
batch, size = (4, 64) if spec == "tiny" else (8, 416)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
out = {}
comm = RcclComm(0, 1, dev, dist.HashStore())
out["rccl"] = comm.rccl_path()
# the primitives: sums over one rank are identities; stream semantics are what matters
t = torch.randn(4099, device=dev)
ref = t.clone()
comm.allreduce_sum(t)
big = torch.randn(1 << 22, device=dev)
bref = big.clone()
big.mul_(2.0)                                   # queued on the compute stream BEFORE the bucket is issued ...
comm.allreduce_async(big)                       # ... the side stream must wait for it
comm.wait_all()
big.mul_(0.5)                                   # ... and the compute stream for the side stream
comm.broadcast(t, 0)
torch.cuda.synchronize()
out["primitives_ok"] = bool(torch.equal(t, ref)) and bool(torch.equal(big, bref))
out["gather"] = comm.all_gather_object({"rank": 0}) == [{"rank": 0}]

clip, head = arch.specs_by_name(spec)
batches = [tuple(x.to(dev) for x in synth.make_batch(batch, size, head.word_len, 0, i)) for i in range(4)]


def run(c, launch):
    tr = NativeTrainer(clip, head, arch.synthetic_state_dict(clip, head, 0), dev, comm=c, sync_bn=c is not None, launch=launch)
    losses = []
    for i in range(steps):
        loss, _ = tr.train_step(*batches[i % 4])
        losses.append(float(loss))
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(20):
        tr.train_step(*batches[i % 4])
    torch.cuda.synchronize()
    return losses, tr.launch, tr.graph_error, (time.time() - t0) / 20 * 1e3, tr.engine.sync_bn


la, launch_a, err_a, ms_a, sb = run(comm, "graph")
lb, launch_b, err_b, ms_b, _ = run(comm, "eager")
from cris.pytorch_amd import debug as _dbg
_dbg.HOOKS.force_dist = False
lc, _, _, ms_c, _ = run(None, "graph")
out.update(sync_bn=sb, launch=launch_a, graph_error=err_a, losses_graph=la, losses_eager=lb, losses_local=lc,
           ms_graph=round(ms_a, 3), ms_eager=round(ms_b, 3), ms_local_graph=round(ms_c, 3))
comm.close()
print("COMM1 " + json.dumps(out))
