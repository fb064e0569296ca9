"""diagnostic: what a device synchronisation per step costs the captured-graph trainer (HIP graph launch latency + pipeline
drain) - the reference's loop synchronises every step (three .item() calls, engine/engine.py:67-69), the native bench loop
never does.   python tools/graph_latency.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import arch, synth                       # noqa: E402
from cris.pytorch_amd.trainer import NativeTrainer              # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
clip, head = arch.specs_by_name("r50")
batches = [tuple(t.to(dev) for t in synth.make_batch(8, 416, head.word_len, 0, i)) for i in range(4)]
for launch in ("graph", "cmdlist"):
    tr = NativeTrainer(clip, head, arch.synthetic_state_dict(clip, head, 0), dev, launch=launch)
    for i in range(6):
        tr.train_step(*batches[i % 4])
    res = {}
    for mode in ("free", "sync", "item"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss, _ = tr.train_step(*batches[i % 4])
            if mode == "sync":
                torch.cuda.synchronize()
            elif mode == "item":
                loss.item()
        torch.cuda.synchronize()
        res[mode] = 1e3 * (time.perf_counter() - t0) / steps
    print("LATENCY launch %s: %.2f ms/step free-running, %.2f with a device sync per step, %.2f with loss.item() per step" % (
        launch, res["free"], res["sync"], res["item"]), flush=True)
    del tr
    torch.cuda.empty_cache()
