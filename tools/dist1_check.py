"""diagnostic: the multi-rank code paths (SyncBN exchanges through torch.distributed, staged gradient all-reduce on its own
communicator and stream) with a ONE-rank RCCL process group on one GPU, in each launch mode - does RCCL work inside the
command list / inside a captured HIP graph with this torch build, and what do the ~145 collectives per step cost in launch
overhead alone?   python tools/dist1_check.py [cmdlist|graph|eager] [steps]"""
import os
import sys
import time

os.environ["CRIS_FORCE_DIST"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import arch, synth                       # noqa: E402
from cris.pytorch_amd.dist import TorchDistComm                 # noqa: E402
from cris.pytorch_amd.trainer import NativeTrainer              # noqa: E402

mode, steps = (sys.argv[1] if len(sys.argv) > 1 else "cmdlist"), int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
comm = TorchDistComm(dev)
clip, head = arch.specs_by_name("r50")
tr = NativeTrainer(clip, head, arch.synthetic_state_dict(clip, head, 0), dev, comm=comm, sync_bn=True, launch=mode)
assert tr.engine.sync_bn
batches = [tuple(t.to(dev) for t in synth.make_batch(8, 416, head.word_len, 0, i)) for i in range(4)]
losses = []
for i in range(6):
    loss, _ = tr.train_step(*batches[i % 4])
    losses.append(float(loss))
torch.cuda.synchronize()
t0 = time.time()
for i in range(steps):
    loss, _ = tr.train_step(*batches[i % 4])
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print("mode %s -> launch %s graph_error %s syncbn %s | %.2f ms/step | losses %s ... %.4f | gradients: %s" % (
    mode, tr.launch, tr.graph_error, tr.syncbn_exchange, dt * 1e3, [round(x, 4) for x in losses], float(loss), tr.grad_exchange), flush=True)
# (the result line is out before the teardown: a watchdog that dies in destroy_process_group cannot take it along)
dist.destroy_process_group()
