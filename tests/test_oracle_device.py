"""The oracle off the CPU: (CPU) the torch restatement of the dropout hash equals the numpy one bit for bit, and the
NATIVE_NORMS switch (F.batch_norm / F.layer_norm instead of the spelled-out forms) does not change the fp32 result beyond
rounding; (GPU) the fp32 oracle on the device equals the pinned CPU run - which keeps the pin when the trajectory tests and
tools/parity_study.py run the oracle on the MI355X."""
import dataclasses

import numpy as np
import pytest
import torch

from cris.pytorch_amd import arch, synth
from oracle import cris_oracle as O
from oracle import dropout_hash


def test_keep_mask_torch_equals_numpy():
    for seed, stream, n, p in ((0, 0, 1000, 0.1), (0xDEADBEEF, 47, 70001, 0.5), (12345, 8 * 2 + 5, 4096, 0.03)):
        a = dropout_hash.keep_mask(seed, stream, n, p)
        b = dropout_hash.keep_mask_torch(seed, stream, n, p, "cpu").numpy()
        assert np.array_equal(a, b)


def _step(sd, clip, head, batch, device, native=False, seed=5):
    O.NATIVE_NORMS = native
    try:
        leaf = {k: (v.to(device).clone().requires_grad_(True) if v.is_floating_point() else v.to(device)) for k, v in sd.items()}
        img, word, mask = (t.to(device) for t in batch)
        bnu = {}
        pred, _, loss = O.cris_forward(leaf, clip, head, img, word, mask, training=True, drop_seed=seed, bn_updates=bnu)
        loss.backward()
        return pred.detach().cpu(), float(loss.detach()), {k: v.grad.cpu() for k, v in leaf.items() if v.requires_grad and v.grad is not None}, bnu
    finally:
        O.NATIVE_NORMS = False


def _tiny():
    clip, head = arch.specs_by_name("tiny")
    head = dataclasses.replace(head, dropout=0.1)
    return clip, head, arch.synthetic_state_dict(clip, head, 0), synth.make_batch(4, 64, head.word_len, 0, 0)


def test_native_norms_same_fp32_result():
    clip, head, sd, batch = _tiny()
    p0, l0, g0, b0 = _step(sd, clip, head, batch, "cpu")
    p1, l1, g1, b1 = _step(sd, clip, head, batch, "cpu", native=True)
    assert abs(l0 - l1) < 1e-5
    assert float((p0 - p1).norm() / p0.norm()) < 1e-4
    worst = min(float(torch.nn.functional.cosine_similarity(g0[k].flatten(), g1[k].flatten(), dim=0)) for k in g0 if g0[k].norm() > 1e-6)      # (k_proj.bias: mathematically zero gradient)
    assert worst > 0.9999
    k = next(iter(b0))
    assert torch.allclose(b0[k][1], b1[k][1], rtol=1e-4, atol=1e-6)          # running_var update: unbiased variance


@pytest.mark.gpu
def test_fp32_oracle_on_the_gpu_equals_the_cpu_run():
    clip, head, sd, batch = _tiny()
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    p0, l0, g0, _ = _step(sd, clip, head, batch, "cpu")
    p1, l1, g1, _ = _step(sd, clip, head, batch, "cuda")
    assert abs(l0 - l1) < 2e-5, (l0, l1)
    assert float((p0 - p1).norm() / p0.norm()) < 2e-4
    cos = [float(torch.nn.functional.cosine_similarity(g0[k].flatten(), g1[k].flatten(), dim=0)) for k in g0 if g0[k].norm() > 1e-6]
    assert min(cos) > 0.999, min(cos)
