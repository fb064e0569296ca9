"""Run the HIP engine and the CPU oracle on the same seeded inputs and report per-tap / per-parameter errors.
Test/diagnostic infrastructure (imports oracle/).  Usage: python tools/parity_report.py [tiny|r50] [B] [S] [dropout]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import arch, synth  # noqa: E402
from cris.pytorch_amd.engine import Engine  # noqa: E402
from oracle import cris_oracle as O  # noqa: E402


def rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def cos(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def nhwc_to_nchw(act):
    t = act.t[:, act.coff:act.coff + act.C].float()
    return t.view(act.Bn, act.H, act.W, act.C).permute(0, 3, 1, 2)


def run(spec="tiny", B=2, S=64, dropout=0.0, seed=11, out=None):
    import dataclasses
    clip, head = arch.specs_by_name(spec)
    head = dataclasses.replace(head, dropout=dropout)
    sd = arch.synthetic_state_dict(clip, head, 0)
    img, word, mask = synth.make_batch(B, S, head.word_len, 0, 0)
    dev = "cuda"
    params = {k: v.to(dev) for k, v in sd.items() if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
    buffers = {k: v.to(dev) for k, v in sd.items() if k.endswith(("running_mean", "running_var"))}
    eng = Engine(clip, head, params, buffers, dev)
    taps = {}
    t0 = time.time()
    pred, msk, loss = eng.forward(img.to(dev), word.to(dev), mask.to(dev), training=True, seed=seed, taps=taps)
    eng.backward()
    G = eng.grads_param_layout()
    torch.cuda.synchronize()
    t_hip = time.time() - t0
    # oracle
    leaf = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    otaps, bnu = {}, {}
    t0 = time.time()
    opred, om, oloss = O.cris_forward(leaf, clip, head, img, word, mask, training=True, drop_seed=seed if dropout > 0 else None,
                                      bn_updates=bnu, taps=otaps)
    oloss.backward()
    t_cpu = time.time() - t0
    # bf16-storage emulation of the oracle (forward only): the noise floor any bf16 implementation shares
    from oracle.bf16_emulation import bf16_storage
    etaps = {}
    with torch.no_grad(), bf16_storage():
        epred, _, eloss = O.cris_forward(sd, clip, head, img, word, mask, training=True, drop_seed=seed if dropout > 0 else None, taps=etaps)
    rep = {"spec": spec, "B": B, "S": S, "dropout": dropout, "loss_hip": float(loss), "loss_oracle": float(oloss),
           "t_hip_s": t_hip, "t_oracle_s": t_cpu, "taps": {}, "grads": {}, "bn": {}}
    rep["mask_equal"] = bool(torch.equal(msk.cpu(), om))
    rep["loss_emul"] = float(eloss)
    rep["emul_vs_fp32"] = {k: rel(etaps[k], otaps[k]) for k in ("layer1", "layer2", "layer3", "layer4", "attnpool", "f5", "fq_neck", "fq_dec")}
    rep["hip_vs_emul"] = {k: rel(nhwc_to_nchw(taps[k]), etaps[k]) for k in ("layer1", "layer2", "layer3", "layer4", "attnpool", "f5", "fq_neck", "fq_dec")}
    rep["hip_vs_emul"]["pred"] = rel(pred, epred)
    rep["emul_vs_fp32"]["pred"] = rel(epred, opred)
    rep["taps"]["pred"] = rel(pred, opred)
    for k in ("layer1", "layer2", "layer3", "layer4", "attnpool", "f5", "f4", "f3", "aggr", "fq_neck", "fq_dec"):
        rep["taps"][k] = rel(nhwc_to_nchw(taps[k]), otaps[k])
    w = taps["word"]
    rep["taps"]["word"] = rel(w.t.float().view(B, -1, w.C), otaps["word"])
    rep["taps"]["state"] = rel(taps["state"].t.float()[:, :taps["state"].C], otaps["state"])
    rep["bn_sorted"] = sorted(((max(rel(buffers[p + ".running_mean"], rm), rel(buffers[p + ".running_var"], rv)), p)
                               for p, (rm, rv) in bnu.items()), reverse=True)[:8]
    worst = []
    for k, g in G.items():
        og = leaf[k].grad
        if og is None:
            rep["grads"][k] = {"oracle_none": True, "hip_norm": float(g.norm())}
            continue
        r, c = rel(g, og), cos(g, og)
        rep["grads"][k] = {"rel": r, "cos": c, "norm": float(og.norm())}
        worst.append((c, r, k))
    worst.sort()
    rep["worst_cos"] = worst[:15]
    for pfx, (rm, rv) in bnu.items():
        rep["bn"][pfx] = [rel(buffers[pfx + ".running_mean"], rm), rel(buffers[pfx + ".running_var"], rv)]
    rep["bn_worst"] = max(max(v) for v in rep["bn"].values())
    if out:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        json.dump(rep, open(out, "w"), indent=1)
    return rep


if __name__ == "__main__":
    spec = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    dp = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    rep = run(spec, B, S, dp, out=os.path.join(ROOT, "gpurun_out", "parity_%s_b%d_s%d_d%g.json" % (spec, B, S, dp)))
    print("loss hip %.6f oracle %.6f | mask_equal %s | t_hip %.2fs" % (rep["loss_hip"], rep["loss_oracle"], rep["mask_equal"], rep["t_hip_s"]))
    print("taps:", {k: "%.2e" % v for k, v in rep["taps"].items()})
    print("loss bf16-emulated oracle %.6f" % rep["loss_emul"])
    print("emul_vs_fp32:", {k: "%.2e" % v for k, v in rep["emul_vs_fp32"].items()})
    print("hip_vs_emul :", {k: "%.2e" % v for k, v in rep["hip_vs_emul"].items()})
    print("bn running stats worst rel err: %.2e" % rep["bn_worst"])
    print("worst bn:", rep["bn_sorted"])
    print("worst grads (cos, rel, name):")
    for c, r, k in rep["worst_cos"]:
        print("   %.4f %.3e %s" % (c, r, k))
    cs = [v["cos"] for v in rep["grads"].values() if "cos" in v]
    print("grad cos: min %.4f median %.4f" % (min(cs), sorted(cs)[len(cs) // 2]))
