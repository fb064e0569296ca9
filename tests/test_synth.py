"""Synthetic RefCOCO-shaped batches (cris.pytorch_amd.synth): the index-level contract the hot path relies on - SOT first,
EOT = the largest id (so `argmax` selects it, reference model/clip.py:451-452), zero padding (pad mask, reference
model/segmenter.py:37), binary masks - and determinism per (rank, step)."""
import torch

from cris.pytorch_amd import synth


def test_token_and_mask_contract():
    img, word, mask = synth.make_batch(8, 64, 17, rank=0, step=3)
    assert img.shape == (8, 3, 64, 64) and img.dtype == torch.float32
    assert word.shape == (8, 17) and word.dtype == torch.int64
    assert mask.shape == (8, 1, 64, 64) and set(mask.unique().tolist()) <= {0.0, 1.0}
    for b in range(8):
        ids = word[b]
        assert int(ids[0]) == synth.SOT
        eot = int(ids.argmax())
        assert int(ids[eot]) == synth.EOT and eot >= 2                 # at least one content token
        assert bool((ids[eot + 1:] == 0).all()) and bool((ids[:eot + 1] != 0).all())
        assert bool(((ids[1:eot] >= 1) & (ids[1:eot] < synth.SOT)).all())


def test_determinism_and_sharding():
    a = synth.make_batch(2, 32, 9, rank=1, step=5)
    b = synth.make_batch(2, 32, 9, rank=1, step=5)
    c = synth.make_batch(2, 32, 9, rank=2, step=5)
    d = synth.make_batch(2, 32, 9, rank=1, step=6)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert not torch.equal(a[0], c[0]) and not torch.equal(a[0], d[0])
