"""Where the bf16 path's loss error comes from: the fp32 oracle runs ON THE GPU as stock PyTorch along its own
training trajectory (the teacher of tests/test_engine_gpu.py's teacher-forced test); at selected states the forward pass is
repeated with bf16 STORAGE rounding (oracle/bf16_emulation.py) switched on for ONE stage of the network at a time, then for ONE
kind of storage point at a time, and the loss is compared with the fp32 loss of the same state, batch and dropout masks.
|dloss| of a group = what that group of storage points contributes by itself; "all" = every point (the floor any bf16
implementation of this network shares); "hip" = the HIP path from the same state.

    python tools/error_budget.py [--spec r50] [--size 416] [--steps 100] [--every 4] [--out gpurun_out/error_budget_r50.json]

Forward passes only (no gradients), ~50 ms each at R50 416x416 batch 8."""
import argparse
import dataclasses
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import arch, synth                          # noqa: E402
from cris.pytorch_amd.trainer import NativeTrainer                # noqa: E402
from oracle import bf16_emulation as E                            # noqa: E402
from oracle import cris_oracle as O                               # noqa: E402
from oracle.torch_runner import OracleTrainer, seed_of_step       # noqa: E402

# groups of storage kinds reported together
KIND_GROUPS = {
    "conv_act": ("conv_in", "conv_out", "relu", "pool", "interp"),     # activations of the convolutional trunk
    "conv_w": ("conv_w",),
    "linear_act": ("linear_in", "linear_out", "ln", "attn_p", "attn_out"),
    "linear_w": ("linear_w",),
}


def emul_loss(ot, batch, seed, kinds=None, stage=None, extra_where=None, skip=None):
    img, word, mask = (t.to(ot.device) for t in batch)
    where = None
    if stage is not None:
        where = (lambda: E.STAGE[0] == stage)
    if extra_where is not None:
        where = extra_where
    with torch.no_grad(), E.staged(), E.bf16_storage(kinds=kinds, where=where, skip=skip):
        _, _, loss = O.cris_forward(ot.leaf, ot.clip, ot.head, img, word, mask, training=True,
                                    drop_seed=seed if ot.head.dropout > 0 else None, bn_updates={})
    return float(loss)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spec", default="r50")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--word-len", type=int, default=None)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--every", type=int, default=4)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--no-hip", action="store_true")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--detail", default=None, help="a stage name: only that stage, one kind of storage point at a time")
    ap.add_argument("--without", default=None, help="stage names joined by '+': the floor with every storage point on EXCEPT that stage's "
                                                    "(all of them / its weights / its activations) - what promoting the stage to fp32 would buy")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device(args.device)
    clip, head = arch.specs_by_name(args.spec)
    head = dataclasses.replace(head, dropout=args.dropout, **({} if args.word_len is None else {"word_len": args.word_len}))
    sd = arch.synthetic_state_dict(clip, head, 0)
    ot = OracleTrainer(clip, head, sd, dev, mode="fp32", lr=args.lr)
    tr = None if args.no_hip else NativeTrainer(clip, head, sd, dev, launch="eager")
    configs = [("all", None, None)] + [("stage:" + s, None, s) for s in E.STAGES] + [("kind:" + k, v, None) for k, v in KIND_GROUPS.items()]
    # the head stages and the text encoder once more, split by kind of storage point (what to promote, if anything)
    configs += [("%s/%s" % (s, k), v, s) for s in ("text", "neck", "decoder", "proj") for k, v in KIND_GROUPS.items()]
    if args.detail:
        configs = [("all", None, None), ("stage:" + args.detail, None, args.detail)]
        configs += [("%s/%s" % (args.detail, k), (k,), args.detail) for k in E.KINDS]
    skips = {}
    if args.without:
        ws = tuple(args.without.split("+"))
        assert all(w in E.STAGES for w in ws), (ws, E.STAGES)
        configs = [("all", None, None)]
        for w in ws:
            skips["all but " + w] = (lambda kind, w=w: E.STAGE[0] == w)
            skips["all but %s weights" % w] = (lambda kind, w=w: E.STAGE[0] == w and kind in ("linear_w", "conv_w"))
            skips["all but %s activations" % w] = (lambda kind, w=w: E.STAGE[0] == w and kind not in ("linear_w", "conv_w"))
        if len(ws) > 1:
            skips["all but " + " + ".join(ws)] = (lambda kind: E.STAGE[0] in ws)
    rows = []
    for t in range(args.steps):
        batch = synth.make_batch(args.batch, args.size, head.word_len, 0, t)
        seed = seed_of_step(t)
        if t % args.every == 0:
            row = {"step": t}
            if tr is not None:
                tr.load_model_state_dict(ot.state_dict())
                img, word, mask = (x.to(dev) for x in batch)
                _, _, loss = tr.engine.forward(img, word, mask, training=True, seed=seed)
                row["hip"] = float(loss)
                tr.engine.tape = []
            with torch.no_grad():
                img, word, mask = (x.to(dev) for x in batch)
                _, _, l32 = O.cris_forward(ot.leaf, clip, head, img, word, mask, training=True,
                                           drop_seed=seed if head.dropout > 0 else None, bn_updates={})
            row["fp32"] = float(l32)
            for name, kinds, stage in configs:
                row[name] = emul_loss(ot, batch, seed, kinds=kinds, stage=stage)
            for name, skip in skips.items():
                row[name] = emul_loss(ot, batch, seed, skip=skip)
            rows.append(row)
            print("BUDGET step %3d fp32 %.5f " % (t, row["fp32"]) + " ".join(
                "%s %.1e" % (k, abs(v - row["fp32"])) for k, v in row.items() if k not in ("step", "fp32")), flush=True)
        if t + 1 < args.steps:
            ot.step(batch, seed)
    names = [k for k in rows[0] if k not in ("step", "fp32")]
    summary = {}
    for k in names:
        d = [abs(r[k] - r["fp32"]) for r in rows]
        sgn = [r[k] - r["fp32"] for r in rows]
        summary[k] = {"mean_abs": sum(d) / len(d), "max_abs": max(d), "mean_signed": sum(sgn) / len(sgn)}
    print("BUDGET summary over %d states (|loss - loss_fp32|: mean / max / mean signed), largest first" % len(rows))
    for k, v in sorted(summary.items(), key=lambda kv: -kv[1]["mean_abs"]):
        print("BUDGET   %-18s %.2e  %.2e  %+.2e" % (k, v["mean_abs"], v["max_abs"], v["mean_signed"]))
    out = args.out or os.path.join(ROOT, "gpurun_out", "error_budget_%s_%d.json" % (args.spec, args.size))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump({"config": vars(args), "rows": rows, "summary": summary}, f)


if __name__ == "__main__":
    main()
