"""diagnostic: where does run-to-run non-determinism enter the backward pass?  Two identical forward+backward runs on one
engine; activation gradients at the taps are compared in backward order."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import arch, synth
from cris.pytorch_amd.engine import Engine, Act

dev = torch.device("cuda:0")
spec, B, S = (sys.argv[1] if len(sys.argv) > 1 else "tiny"), int(sys.argv[2]) if len(sys.argv) > 2 else 4, int(sys.argv[3]) if len(sys.argv) > 3 else 64
clip, head = arch.specs_by_name(spec)
sd = arch.synthetic_state_dict(clip, head, 0)
params = {k: v.to(dev) for k, v in sd.items() if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
buffers = {k: v.to(dev) for k, v in sd.items() if k.endswith(("running_mean", "running_var"))}
e = Engine(clip, head, params, buffers, dev)
if os.environ.get("CRIS_NO_SIDE") == "1":
    e.side = None
img, word, mask = (t.to(dev) for t in synth.make_batch(B, S, head.word_len, 0, 0))

def once():
    taps = {}
    e.forward(img, word, mask, training=True, seed=17, taps=taps)
    e.backward()
    torch.cuda.synchronize()
    acts = {k: v.t.clone() for k, v in taps.items() if isinstance(v, Act)}
    grads = {k: v.g.clone() for k, v in taps.items() if isinstance(v, Act) and v.g is not None}
    return acts, grads, {k: v.clone() for k, v in e.grads_param_layout().items()}

a1, g1, p1 = once()
a2, g2, p2 = once()
rel = lambda x, y: float((x.float() - y.float()).norm() / (y.float().norm() + 1e-30))
print("forward taps equal:", {k: bool(torch.equal(a1[k], a2[k])) for k in a1})
for k in ("fq_dec", "fq_neck", "aggr", "f3", "f4", "f5", "s", "state", "word", "attnpool", "layer4", "layer3", "layer2", "layer1"):
    if k in g1:
        print("d%-9s run-to-run rel diff %.3e  equal %s" % (k, rel(g1[k], g2[k]), bool(torch.equal(g1[k], g2[k]))))
w = sorted(((rel(p1[k], p2[k]), k) for k in p1), reverse=True)
print("param grads worst:", [(round(a, 5), b) for a, b in w[:6]])
print("param grads bit-identical: %d of %d" % (sum(bool(torch.equal(p1[k], p2[k])) for k in p1), len(p1)))
