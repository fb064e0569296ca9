"""Generate golden fixtures by RUNNING THE REFERENCE ITSELF (this container only).

    python tests/golden/make_golden.py

Imports DerrickWang005/CRIS.pytorch from /root/reference (read-only) via ref_harness.py, loads the
deterministic synthetic state_dict of cris.pytorch_amd.arch into `model.segmenter.CRIS`, runs
train-mode forward + backward (dropout 0 - torch's Philox dropout stream cannot be reproduced by any
other implementation) and eval-mode forward on seeded synthetic batches, and stores the results as
small .npz fixtures next to this script.  tests/test_oracle_golden.py pins oracle/cris_oracle.py
against them; the GPU parity tests then compare the HIP path with the oracle and with these files.
Fixtures hold outputs only (weights/inputs are regenerated from seeds on any machine).
"""
import dataclasses
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from cris.pytorch_amd import arch, synth  # noqa: E402
import ref_harness  # noqa: E402

CASES = {
    # name: (spec, batch, size, dropout-free)
    "tiny_b2_s64": ("tiny", 2, 64),
    "tiny_b3_s96": ("tiny", 3, 96),
    "r50_b2_s160": ("r50", 2, 160),
    "r101_b2_s96": ("r101", 2, 96),              # BASELINE.json configs[3] parameter tree (layer3 x23, embed_dim 512)
    "tiny_b2_s96_l22": ("tiny", 2, 96, 22),      # configs[4]'s longer expressions (word_len 22), odd 3x3 / 6x6 / 12x12 maps
}


def cfg_from(head, **over):
    d = {k: (list(v) if isinstance(v, tuple) else v) for k, v in dataclasses.asdict(head).items()}
    d.update(clip_pretrain="synthetic", **over)
    return d


def run_case(name, spec, batch, size, word_len=None):
    clip, head = arch.specs_by_name(spec)
    if word_len is not None:
        head = dataclasses.replace(head, word_len=word_len)
    sd = arch.synthetic_state_dict(clip, head, seed=0)
    net = ref_harness.build_reference_cris(arch.clip_state_dict_view(sd), cfg_from(head, dropout=0.0))
    missing = net.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    img, word, mask = synth.make_batch(batch, size, head.word_len, rank=0, step=0)
    out = {}
    # eval forward (running statistics)
    net.eval()
    with torch.no_grad():
        out["eval_pred"] = net(img, word).numpy()
    # train forward + backward
    net.train()
    rm_before = {k: v.clone() for k, v in net.state_dict().items() if k.endswith("running_mean")}
    pred, m, loss = net(img, word, mask)
    loss.backward()
    out["train_pred"] = pred.numpy()
    out["train_mask"] = m.numpy()
    out["loss"] = np.array(loss.item(), dtype=np.float64)
    names, norms, sums = [], [], []
    samples = {}
    for k, p in net.named_parameters():
        if p.grad is None:
            names.append(k); norms.append(-1.0); sums.append(0.0)
            continue
        g = p.grad.double()
        names.append(k); norms.append(float(g.norm())); sums.append(float(g.sum()))
        flat = p.grad.flatten()
        n = min(64, flat.numel())
        idx = (torch.arange(n, dtype=torch.int64) * (flat.numel() - 1)) // max(n - 1, 1)
        samples[k] = flat[idx].numpy()
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms)
    out["grad_sums"] = np.array(sums)
    keep = [k for k in samples if ("layers.0" in k or "proj" in k or "conv1" in k or "embedding" in k
                                   or "txt_proj" in k or "attnpool" in k or "ln_final" in k or "coordconv" in k)]
    for k in keep:
        out["gs:" + k] = samples[k]
    # BN running statistics after one train step (a few layers)
    after = net.state_dict()
    for k in list(rm_before)[:6] + list(rm_before)[-3:]:
        out["rm:" + k] = after[k].numpy()
        out["rv:" + k] = after[k.replace("running_mean", "running_var")].numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", loss.item(), "params", len(names), "file KB",
          os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024)


def tokenizer_vectors():
    """Known answers of the reference tokenizer (utils/dataset.py:43-84 + utils/simple_tokenizer.py)
    with the ftfy stub - the bit-exact contract for token ids / EOT argmax."""
    ref_harness.import_reference()
    from utils.dataset import tokenize
    cases = [("a diagram", 17), ("the man in the red shirt on the left", 17),
             ("woman holding an umbrella standing behind the bench near the big tree with a dog and a cat and a bird", 17),
             ("giraffe's head", 22)]
    out = []
    for text, L in cases:
        ids = tokenize(text, L, True)[0]
        out.append({"text": text, "context_length": L, "ids": ids.tolist(), "argmax": int(ids.argmax()),
                    "pad_mask": (ids == 0).int().tolist()})
    with open(os.path.join(HERE, "tokenizer_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("tokenizer vectors", len(out))


def metric_vectors():
    """Known answers of the reference's train metric (utils/misc.py:114-129 trainMetricGPU) on seeded logits / masks,
    including logits exactly at the 0.35 threshold and empty masks (union 0 -> IoU 0 through the +1e-6)."""
    ref_harness.import_reference()
    from utils.misc import trainMetricGPU
    out = {}
    thr_logit = float(torch.log(torch.tensor(0.35 / 0.65)))
    for i, (b, hw) in enumerate([(4, 26), (8, 104), (3, 13), (2, 8)]):
        g = torch.Generator().manual_seed(500 + i)
        pred = torch.randn(b, 1, hw, hw, generator=g) * 2.0
        target = (torch.rand(b, 1, hw, hw, generator=g) > 0.6).float()
        if i == 2:
            target[0] = 0.0                                   # empty ground truth
            pred[0] = -10.0                                   # and empty prediction: union 0
        if i == 3:
            pred[0, 0, 0, :4] = thr_logit                     # sigmoid == threshold up to rounding
        iou, prec = trainMetricGPU(pred.clone(), target.clone())
        out["iou_%d" % i] = np.array(float(iou), dtype=np.float64)
        out["prec_%d" % i] = np.array(float(prec), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "train_metric.npz"), **out)
    print("train metric vectors", len(out) // 2)


def key_listing():
    clip, head = arch.specs_by_name("r50")
    sd = arch.synthetic_state_dict(clip, head, seed=0)
    net = ref_harness.build_reference_cris(arch.clip_state_dict_view(sd), cfg_from(head))
    with open(os.path.join(HERE, "state_dict_keys_r50.txt"), "w") as f:
        for k, v in net.state_dict().items():
            f.write("%s %s %s\n" % (k, tuple(v.shape), str(v.dtype).replace("torch.", "")))
    g0 = [k for k, _ in net.named_parameters() if k.startswith("backbone") and "positional_embedding" not in k]
    print("keys", len(net.state_dict()), "group A", len(g0))


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if not sys.argv[1:]:
        key_listing()
        tokenizer_vectors()
    if not sys.argv[1:] or "train_metric" in sys.argv[1:]:
        metric_vectors()
    only = sys.argv[1:]
    for name, case in CASES.items():
        if not only or name in only:
            run_case(name, *case)
