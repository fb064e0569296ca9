"""Parity by measurement: the oracle run ON THE MI355X as stock PyTorch, beside the HIP path, all from the
same state, batches and dropout masks (oracle/torch_runner.py):

  (a) fp32 oracle on the GPU  - the reference trajectory (equal to the pinned CPU oracle: tests/test_oracle_device.py, and
      compared here with the committed CPU fixture tests/golden/traj_r50_...json);
  (b) torch.autocast(float16) + GradScaler - the REFERENCE's own precision policy (engine/engine.py:48-57);
  (c) torch.autocast(bfloat16);
  (d) the HIP path (NativeTrainer, captured graph).

Reports, for BASELINE.json configs[1] (R50, 416x416, batch 8, dropout 0.1, Adam lr 1e-4, 100 steps): |loss - loss_fp32| per
phase of the free-running trajectories, step-0 logits error and per-tensor gradient cosines against fp32 (median / worst)
for (b), (c), (d), and - a diagnostic, not a target - the ms/step of stock PyTorch-ROCm on the same model and GPU.

    python tools/parity_study.py [--steps 100] [--out gpurun_out/parity_r03.json]
"""
import argparse
import dataclasses
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import arch, synth                          # noqa: E402
from cris.pytorch_amd.trainer import NativeTrainer                # noqa: E402
from oracle.torch_runner import OracleTrainer, cosines, seed_of_step   # noqa: E402

PHASES = [(0, 5), (5, 40), (40, 100), (60, 100)]


def phase_stats(d):
    out = {}
    for lo, hi in PHASES:
        seg = d[lo:hi]
        if seg:
            out["%d-%d" % (lo, hi - 1)] = {"max": max(seg), "mean": sum(seg) / len(seg)}
    return out


def med(d):
    v = sorted(d.values())
    return v[len(v) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spec", default="r50")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--modes", default="fp32,fp16,bf16,hip")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_r03.json"))
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    clip, head = arch.specs_by_name(args.spec)
    head = dataclasses.replace(head, dropout=args.dropout)
    sd = arch.synthetic_state_dict(clip, head, 0)
    batches = [synth.make_batch(args.batch, args.size, head.word_len, 0, t) for t in range(args.steps)]
    rep = {"config": vars(args), "loss": {}, "ms_per_step": {}, "step0": {}}
    modes = args.modes.split(",")
    g_fp32 = p_fp32 = None
    for mode in modes:
        torch.cuda.empty_cache()
        t_steps = []
        if mode == "hip":
            tr = NativeTrainer(clip, head, sd, dev, base_lr=args.lr, launch="eager")
            e = tr.engine
            img, word, mask = (t.to(dev) for t in batches[0])
            pred, _, loss0 = e.forward(img, word, mask, training=True, seed=seed_of_step(0))
            e.backward()
            torch.cuda.synchronize()
            g0 = {k: v.detach().clone() for k, v in e.grads_param_layout().items()}
            p0 = pred.detach().float().clone()
            del tr, e
            tr = NativeTrainer(clip, head, sd, dev, base_lr=args.lr)
            losses = []
            for t in range(args.steps):
                img, word, mask = (x.to(dev) for x in batches[t])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                loss, _ = tr.train_step(img, word, mask)
                losses.append(float(loss))
                t_steps.append(time.perf_counter() - t0)
            del tr
        else:
            ot = OracleTrainer(clip, head, sd, dev, mode=mode, lr=args.lr)
            losses = []
            for t in range(args.steps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                loss, pred = ot.forward_backward(batches[t], seed_of_step(t))
                if t == 0:
                    g0 = {k: v.clone() for k, v in ot.grads().items()}
                    p0 = pred.clone()
                ot.update()
                torch.cuda.synchronize()
                t_steps.append(time.perf_counter() - t0)
                losses.append(loss)
            del ot
        rep["loss"][mode] = losses
        tail = sorted(t_steps[len(t_steps) // 2:])
        rep["ms_per_step"][mode] = 1e3 * tail[len(tail) // 2]
        if mode == "fp32":
            g_fp32, p_fp32 = g0, p0
        elif g_fp32 is not None:
            cs = cosines(g0, g_fp32)
            worst = min(cs, key=cs.get)
            rep["step0"][mode] = {"loss_diff": abs(losses[0] - rep["loss"]["fp32"][0]),
                                  "logits_rel_l2": float((p0 - p_fp32).norm() / p_fp32.norm()),
                                  "grad_cos_median": med(cs), "grad_cos_worst": cs[worst], "grad_cos_worst_name": worst,
                                  "grad_cos_below_0.95": sum(1 for v in cs.values() if v < 0.95), "n": len(cs)}
        print("PARITY mode %s: %.1f ms/step (median of the second half; torch modes include the hash-mask dropout in torch ops), "
              "loss[0..4] %s ... final %.4f" % (mode, rep["ms_per_step"][mode], ["%.4f" % x for x in losses[:5]], losses[-1]), flush=True)
    if "fp32" in rep["loss"]:
        ref = rep["loss"]["fp32"]
        rep["abs_loss_diff_vs_fp32"] = {m: phase_stats([abs(a - b) for a, b in zip(rep["loss"][m], ref)]) for m in rep["loss"] if m != "fp32"}
        fx = os.path.join(ROOT, "tests", "golden", "traj_%s_b%d_s%d_d%g_lr%g.json" % (args.spec, args.batch, args.size, args.dropout, args.lr))
        if os.path.exists(fx):
            cpu = json.load(open(fx))["loss"][:len(ref)]
            rep["abs_loss_diff_vs_fp32"]["fp32_cpu_fixture"] = phase_stats([abs(a - b) for a, b in zip(cpu, ref)])
        for m, st in rep["abs_loss_diff_vs_fp32"].items():
            print("PARITY |loss - fp32(gpu)| %-18s %s" % (m, "  ".join("%s: max %.2e mean %.2e" % (k, v["max"], v["mean"]) for k, v in st.items())))
    for m, st in rep["step0"].items():
        print("PARITY step 0 vs fp32 %-5s %s" % (m, {k: (round(v, 5) if isinstance(v, float) else v) for k, v in st.items()}))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rep, f)


if __name__ == "__main__":
    main()
