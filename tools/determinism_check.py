"""diagnostic: two trainers from the same state, the same batches, N steps - the first step whose loss differs, and (eager)
which gradients differ there.   python tools/determinism_check.py r50 8 416 12 [graph|eager]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import arch, synth
from cris.pytorch_amd.trainer import NativeTrainer

spec, B, S, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
launch = sys.argv[5] if len(sys.argv) > 5 else "graph"
dev = torch.device("cuda:0")
clip, head = arch.specs_by_name(spec)
runs = []
for r in range(2):
    sd = arch.synthetic_state_dict(clip, head, 0)
    tr = NativeTrainer(clip, head, sd, dev, launch=launch)
    losses, arenas = [], []
    for t in range(N):
        img, word, mask = synth.make_batch(B, S, head.word_len, 0, t % 4)
        loss, _ = tr.train_step(img.to(dev), word.to(dev), mask.to(dev))
        torch.cuda.synchronize()
        losses.append(float(loss))
        if launch == "eager":
            arenas.append({k: v.clone() for k, v in tr.engine.G.items()})
    runs.append((losses, arenas, {k: v.clone() for k, v in tr.engine.P.items()}))
    del tr
(l0, a0, p0), (l1, a1, p1) = runs
first = next((i for i, (x, y) in enumerate(zip(l0, l1)) if x != y), None)
print(spec, B, S, launch, "env", {k: v for k, v in os.environ.items() if k.startswith("CRIS_")})
print("losses run0", [round(x, 6) for x in l0])
print("losses run1", [round(x, 6) for x in l1])
print("first differing step:", first, "| params identical at the end:", all(torch.equal(p0[k], p1[k]) for k in p0))
if launch == "eager":
    for t in range(N):
        bad = [k for k in a0[t] if not torch.equal(a0[t][k], a1[t][k])]
        if bad:
            print("step", t, "gradients that differ (%d of %d):" % (len(bad), len(a0[t])), bad[:12])
            break
