"""Inference path (cris/pytorch_amd/infer.py): the eval-mode forward with the BatchNorms folded into their
convolutions and replayed as a HIP graph, against (a) the same engine without folding, (b) the CPU oracle's eval forward
(oracle/cris_oracle.py, pinned to the reference's own outputs by tests/test_oracle_golden.py), and against itself: graph
replays equal eager launches bit for bit.  Bounds are fixed numbers (bf16 operands; folding moves one rounding from the
accumulator side to the weight side)."""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu

from cris.pytorch_amd import arch, evalpost, synth  # noqa: E402
from cris.pytorch_amd.infer import InferenceRunner  # noqa: E402
from oracle import cris_oracle as O  # noqa: E402

DEV = torch.device("cuda:0")


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("spec,batch,size,word_len", [("tiny", 3, 96, 9), ("r50", 1, 416, 17)])
def test_folded_forward_matches_unfolded_and_oracle(spec, batch, size, word_len):
    clip, head = arch.specs_by_name(spec)
    head = dataclasses.replace(head, word_len=word_len)
    sd = arch.synthetic_state_dict(clip, head, 0)
    img, word, _ = synth.make_batch(batch, size, word_len, 0, 0)
    folded = InferenceRunner(clip, head, sd, DEV, fold_bn=True, use_graph=False)
    plain = InferenceRunner(clip, head, sd, DEV, fold_bn=False, use_graph=False)
    a = folded(img.to(DEV), word.to(DEV)).clone()
    b = plain(img.to(DEV), word.to(DEV)).clone()
    torch.cuda.synchronize()
    e = folded.engine
    # every BatchNorm but neck.f1_v_proj.1 / neck.norm_layer.0 (a per-sample multiplication sits between them) and the BatchNorm1d
    # of neck.txt_proj, which runs in fp32 on the sentence vector's rows (csrc/smallf32.hip) when that path is on
    assert len(e._fold) == len(e.bn_prefixes) - (3 if (e.state_f32 and batch <= 16) else 2)
    assert len(plain.engine._fold) == 0
    with torch.no_grad():
        ref = O.cris_forward(sd, clip, head, img, word, training=False)
    assert a.shape == ref.shape == (batch, 1, size // 4, size // 4)
    ef, ep, eo = _rel(a.cpu(), b.cpu()), _rel(b.cpu(), ref), _rel(a.cpu(), ref)
    print("folded vs unfolded %.3e | unfolded vs oracle %.3e | folded vs oracle %.3e" % (ef, ep, eo))
    # measured (call AB): tiny 5.9e-3 / 8.6e-3 / 7.3e-3; R50 416x416 batch 1: 1.9e-2 / 1.7e-2 / 9.2e-3 (the two HIP paths sit on
    # different sides of the oracle; results are bit-reproducible, so these are fixed numbers, not noise margins)
    assert ef < 3e-2 and eo < 2e-2, (ef, ep, eo)


def test_graph_replay_equals_eager_and_upsample_is_fused():
    """calls 1 (eager), 2 (captured), 3.. (replayed) on different batches == a runner that launches every kernel from Python;
    with upsample=True the runner returns sigmoid + bicubic (align_corners) probabilities at the input size"""
    clip, head = arch.specs_by_name("tiny")
    sd = arch.synthetic_state_dict(clip, head, 0)
    g = InferenceRunner(clip, head, sd, DEV, use_graph=True)
    e = InferenceRunner(clip, head, sd, DEV, use_graph=False)
    u = InferenceRunner(clip, head, sd, DEV, use_graph=True, upsample=True)
    for t in range(5):
        img, word, _ = synth.make_batch(4, 64, head.word_len, 0, t)
        img, word = img.to(DEV), word.to(DEV)
        a, b = g(img, word).clone(), e(img, word).clone()
        p = u(img, word).clone()
        torch.cuda.synchronize()
        assert torch.equal(a, b), t
        assert p.shape == (4, 64, 64) and torch.equal(p, evalpost.sigmoid_upsample(b, 64, 64))
    assert g.graph_error is None and u.graph_error is None
    assert next(iter(g._shapes.values()))["graph"] is not None


def test_load_state_dict_refolds():
    clip, head = arch.specs_by_name("tiny")
    sd0, sd1 = arch.synthetic_state_dict(clip, head, 0), arch.synthetic_state_dict(clip, head, 1)
    img, word, _ = synth.make_batch(2, 64, head.word_len, 0, 0)
    img, word = img.to(DEV), word.to(DEV)
    r = InferenceRunner(clip, head, sd0, DEV)
    for _ in range(3):
        a0 = r(img, word).clone()
    r.load_state_dict(sd1)
    for _ in range(3):
        a1 = r(img, word).clone()
    fresh = InferenceRunner(clip, head, sd1, DEV, use_graph=False)(img, word)
    torch.cuda.synchronize()
    assert torch.equal(a1, fresh) and not torch.equal(a0, a1)
