"""Replay a loss-trajectory fixture (tests/golden/traj_*.json, made by the CPU oracle + torch Adam) on the HIP trainer
and print both curves.  usage: python tools/trajectory_check.py traj_r50_b8_s416_d0.1_lr2e-06.json [max_steps]"""
import dataclasses
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import arch, synth  # noqa: E402
from cris.pytorch_amd.trainer import NativeTrainer  # noqa: E402


def run(name, max_steps=None):
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", name)))
    clip, head = arch.specs_by_name(fx["spec"])
    head = dataclasses.replace(head, dropout=fx["dropout"])
    sd = arch.synthetic_state_dict(clip, head, 0)
    dev = torch.device("cuda:0")
    tr = NativeTrainer(clip, head, sd, dev, base_lr=fx["lr"])
    n = len(fx["loss"]) if max_steps is None else min(max_steps, len(fx["loss"]))
    out = []
    for t in range(n):
        img, word, mask = synth.make_batch(fx["batch"], fx["size"], head.word_len, 0, t)
        loss, metric = tr.train_step(img.to(dev), word.to(dev), mask.to(dev))
        out.append((float(loss), fx["loss"][t], float(metric[0]), fx["iou"][t]))
    return out


if __name__ == "__main__":
    rows = run(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
    print("step  hip_loss  oracle_loss   |d|      hip_iou  oracle_iou")
    for t, (a, b, c, d) in enumerate(rows):
        print("%4d  %.5f   %.5f   %.2e   %7.3f  %7.3f" % (t, a, b, abs(a - b), c, d))
    ds = [abs(a - b) for a, b, _, _ in rows]
    print("max |d| %.3e  mean |d| %.3e over %d steps" % (max(ds), sum(ds) / len(ds), len(ds)))
