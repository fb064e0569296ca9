"""Generate loss-trajectory fixtures with the CPU oracle + torch.optim.Adam (this container; ~1.5 h of CPU for R50 at full size).

    python tests/golden/make_trajectory.py r50 8 416 100 0.1 1e-4        -> tests/golden/traj_r50_b8_s416_d0.1_lr0.0001.json
    python tests/golden/make_trajectory.py r50 8 416 100 0.1 1e-4 emul   -> ..._bf16emul.json (bf16 storage rounding, informative)

Protocol = the reference train loop on synthetic data (engine/engine.py:37-57, train.py:105-107): fresh seeded batch per
step (synth.make_batch(rank 0, step t)), forward + BCE loss, backward, Adam(lr, betas .9/.999, eps 1e-8, wd 0) over
every parameter that receives a gradient, BatchNorm running statistics updated.  lr = the reference's 1e-4
(config/refcoco/cris_r50.yaml) for every fixture.  With the untrained synthetic head the first ~30 steps are violent for any
implementation (fp32 loss 0.90, 1.80, 1.46, 0.78, 1.11, 2.17, ...), then the curve settles.  (A much smaller lr is NOT a
gentler test for a bf16 path: the CLIP weights sit on the fp16 grid, 1/8 of them exactly on a bf16 rounding tie, and a
+-2e-6 Adam step flips every one of those ties to the descent side - a coherent half-ulp move ~4x the step itself; measured
with this oracle: loss after one step 0.729 in fp32 against 0.377 with bf16 weight rounding, 0.753 once the weights are
dithered off the grid.)  Dropout masks come from the shared counter hash (oracle/dropout_hash.py) with the trainer's seed
rule (step*7919+17), so the HIP path can be compared step by step at the full BASELINE.json configs[1] shape WITHOUT running
the CPU oracle on the GPU box.  The file is rewritten after every step (partial runs are usable)."""
import dataclasses
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from cris.pytorch_amd import arch, synth  # noqa: E402
from oracle import cris_oracle as O  # noqa: E402


def main(spec="tiny", B=4, S=64, steps=100, dropout=0.1, lr=1e-4, emul=False, threads=4):
    torch.set_num_threads(threads)
    clip, head = arch.specs_by_name(spec)
    head = dataclasses.replace(head, dropout=dropout)
    sd = arch.synthetic_state_dict(clip, head, 0)
    leaf = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and not k.endswith(("running_mean", "running_var")))
                else v.clone()) for k, v in sd.items()}
    params = [v for v in leaf.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=lr)
    out = os.path.join(HERE, "traj_%s_b%d_s%d_d%g_lr%g%s.json" % (spec, B, S, dropout, lr, "_bf16emul" if emul else ""))
    import contextlib
    from oracle.bf16_emulation import bf16_storage
    ctx = bf16_storage if emul else contextlib.nullcontext
    rec = {"spec": spec, "batch": B, "size": S, "dropout": dropout, "lr": lr, "seed_rule": "step*7919+17", "bf16_storage_emulation": emul,
           "loss": [], "iou": [], "pr50": [], "sec_per_step": []}
    for t in range(steps):
        t0 = time.time()
        img, word, mask = synth.make_batch(B, S, head.word_len, 0, t)
        bnu = {}
        with ctx():
            pred, m, loss = O.cris_forward(leaf, clip, head, img, word, mask, training=True,
                                           drop_seed=(t * 7919 + 17) if dropout > 0 else None, bn_updates=bnu)
            opt.zero_grad()
            loss.backward()
        opt.step()
        with torch.no_grad():
            for pfx, (rm, rv) in bnu.items():
                leaf[pfx + ".running_mean"].copy_(rm)
                leaf[pfx + ".running_var"].copy_(rv)
            iou, pr = O.train_metric(pred.detach(), m)
        rec["loss"].append(float(loss.detach()))
        rec["iou"].append(float(iou))
        rec["pr50"].append(float(pr))
        rec["sec_per_step"].append(round(time.time() - t0, 2))
        with open(out, "w") as f:
            json.dump(rec, f)
        print(t, rec["loss"][-1], rec["sec_per_step"][-1], flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0] if a else "tiny", int(a[1]) if len(a) > 1 else 4, int(a[2]) if len(a) > 2 else 64,
         int(a[3]) if len(a) > 3 else 100, float(a[4]) if len(a) > 4 else 0.1, float(a[5]) if len(a) > 5 else 1e-4,
         emul=len(a) > 6 and a[6] == "emul")
