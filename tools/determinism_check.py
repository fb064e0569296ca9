"""diagnostic: two trainers from the same state, the same batches, N steps - the first step whose loss differs, and (eager)
which gradients differ there.   python tools/determinism_check.py r50 8 416 12 [graph|eager]
Environment: CRIS_DEBUG=sleep holds the device back at the backward fork (text and visual backward then run fully
concurrently - the worst case for cross-stream interference); TAPS=1 (eager) also records stream-ordered copies of the text
encoder's gradient buffers after every backward closure and names the first closure whose output differs, with a
decomposition of a LayerNorm-backward difference; CRIS_NO_SIDE=1 puts the text encoder on the launch stream."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import arch, debug, synth
from cris.pytorch_amd.trainer import NativeTrainer

spec, B, S, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
launch = sys.argv[5] if len(sys.argv) > 5 else "graph"
dev = torch.device("cuda:0")
clip, head = arch.specs_by_name(spec)
runs = []
for r in range(2):
    sd = arch.synthetic_state_dict(clip, head, 0)
    tr = NativeTrainer(clip, head, sd, dev, launch=launch)
    losses, arenas, taps = [], [], []
    for t in range(N):
        img, word, mask = synth.make_batch(B, S, head.word_len, 0, t % 4)
        if os.environ.get("TAPS") == "1":
            debug.HOOKS.taps = []
        loss, _ = tr.train_step(img.to(dev), word.to(dev), mask.to(dev))
        torch.cuda.synchronize()
        if os.environ.get("TAPS") == "1":
            taps.append(debug.HOOKS.taps)
        losses.append(float(loss))
        if launch == "eager":
            arenas.append({k: v.clone() for k, v in tr.engine.G.items()})
    runs.append((losses, arenas, {k: v.clone() for k, v in tr.engine.P.items()}, taps))
    del tr
(l0, a0, p0, t0), (l1, a1, p1, t1) = runs
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
            if t0:
                shown, prev, analysed = 0, None, False
                for ta, tb in zip(t0[t], t1[t]):
                    (i, name, ra, ea), (_, _, rb, eb) = ta, tb
                    for (ka, xa), (kb, xb) in zip(ra, rb):
                        if not torch.equal(xa, xb):
                            d = (xa.float() - xb.float()).abs().flatten()
                            idx = torch.nonzero(d > 0).flatten()
                            print("   TAP closure %d %s %s shape %s diff %d [%d..%d] max|d| %.3e max|x| %.3e" % (
                                i, name, ka, tuple(xa.shape), idx.numel(), int(idx[0]), int(idx[-1]), float(d.max()), float(xa.float().abs().max())))
                            shown += 1
                            if not analysed and name.startswith("Engine.ln.") and "dx_stream.g" in ea:
                                analysed = True
                                for nm in ea:
                                    print("      input %-14s equal between runs: %s" % (nm, torch.equal(ea[nm], eb[nm])))
                                # recompute the kernel's dx for the differing rows from the tapped inputs (fp32 torch)
                                rows = torch.unique(idx // xa.shape[1]).tolist()
                                pa = {k2: v2 for k2, v2 in prev[3].items()} if prev is not None else {}
                                x, mean, rstd, gamma = ea["x.t"].float(), ea["mean"], ea["rstd"], ea["gamma"].float()
                                dy = ea["y.g"].float()
                                xh = (x - mean[:, None]) * rstd[:, None]
                                a = dy * gamma[None, :]
                                o = rstd[:, None] * (a - a.mean(1, keepdim=True) - xh * (a * xh).mean(1, keepdim=True))
                                for r in rows[:4]:
                                    da, db = ea["dx_stream.g"][r], eb["dx_stream.g"][r]
                                    dd = (da - db)
                                    basis = torch.stack([o[r], xh[r], torch.ones_like(xh[r])], 1)
                                    sol = torch.linalg.lstsq(basis, dd[:, None]).solution.flatten()
                                    res = dd - basis @ sol
                                    print("      row %d: |dd| %.3e coeffs(o, xh, 1) %s residual %.3e spikes %d  mean %.6f rstd %.6f" % (
                                        r, float(dd.norm()), [float("%.3e" % c) for c in sol], float(res.norm()),
                                        int((res.abs() > 5 * res.abs().mean()).sum()), float(mean[r]), float(rstd[r])))
                    prev = ta
                    if shown >= 8:
                        break
            for k in bad[:6]:
                x, y = a0[t][k].float().flatten(), a1[t][k].float().flatten()
                d = (x - y).abs()
                idx = torch.nonzero(d > 0).flatten()
                print("   %-62s n %8d diff %8d [%d..%d] max|d| %.3e max|x| %.3e" % (k, x.numel(), idx.numel(), int(idx[0]), int(idx[-1]),
                                                                                      float(d.max()), float(x.abs().max())))
            break
