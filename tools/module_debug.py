"""diagnostic: drop-in module (torch Adam + GradScaler) vs NativeTrainer on the same step"""
import os, sys
from types import SimpleNamespace as NS
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cris.pytorch_amd import arch, synth
from cris.pytorch_amd.model import build_segmenter
from cris.pytorch_amd.trainer import NativeTrainer
from test_module_surface import TINY

dev = torch.device("cuda:0")
model, groups = build_segmenter(NS(**TINY)); model = model.to(dev).train()
clip, head = arch.specs_by_name("tiny")
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
tr = NativeTrainer(clip, head, sd, dev, base_lr=1e-4, use_graph=False)
img, word, mask = (t.to(dev) for t in synth.make_batch(4, 64, 9, 0, 0))
for use_scaler in (False, True):
    model.zero_grad()
    model._steps = 0
    scale = 65536.0 if use_scaler else 1.0
    with torch.autocast("cuda"):
        pred, target, loss = model(img, word, mask)
    (loss * scale).backward()
    e = tr.engine
    p2, m2, l2 = e.forward(img, word, mask, training=True, seed=17)
    e.backward()
    G = e.grads_param_layout()
    worst = []
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        a, b = p.grad.float() / scale, G[n].float()
        d = float((a - b).norm() / (b.norm() + 1e-30))
        worst.append((d, n, bool(torch.isfinite(p.grad).all())))
    worst.sort(reverse=True)
    Gm = model._engine.grads_param_layout()
    exp = sorted(((float((p.grad.float() / scale - Gm[n].float() / scale).norm() / (Gm[n].float().norm() / scale + 1e-30)), n)
                  for n, p in model.named_parameters() if p.grad is not None), reverse=True)
    print("  export (p.grad vs module engine arena) worst:", exp[:3])
    eng = sorted(((float((Gm[n].float() / scale - G[n].float()).norm() / (G[n].float().norm() + 1e-30)), n) for n in G if n in Gm), reverse=True)
    print("  module engine arena vs trainer engine arena worst:", eng[:4])
    print("scaler", use_scaler, "loss", float(loss), float(l2), "pred equal", bool(torch.equal(pred, p2)))
    print("  worst grad rel diff:", worst[:6])
    print("  all finite:", all(w[2] for w in worst))

# ---- run-to-run determinism of the engine's backward (same engine, same inputs, same seed)
def grads_once():
    e.forward(img, word, mask, training=True, seed=17)
    e.backward()
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in e.grads_param_layout().items()}
g1, g2 = grads_once(), grads_once()
w = sorted(((float((g1[k] - g2[k]).norm() / (g2[k].norm() + 1e-30)), k) for k in g1), reverse=True)
print("run-to-run grad rel diff (same engine):", w[:8])
