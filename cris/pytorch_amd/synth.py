"""Synthetic RefCOCO-shaped batches (there is no network for datasets).

Mirrors what reference utils/dataset.py:154-168 hands the train loop - (img f32 [3,S,S] already
mean/std normalised, word int64 [L] = SOT ids... EOT 0-padded, mask f32 [S,S] in {0,1}) - and the
dummy tensors of reference tools/latency.py:51-52.  Seeding rule from SURVEY.md section 8(d):
generator seed = 1234 + 1000*rank + step.
"""
import torch

SOT, EOT = 49406, 49407


def make_batch(batch, size, word_len, rank=0, step=0, vocab=49408, rect_mask=True):
    g = torch.Generator(device="cpu")
    g.manual_seed(1234 + 1000 * rank + step)
    img = torch.randn(batch, 3, size, size, generator=g)
    word = torch.zeros(batch, word_len, dtype=torch.int64)
    hi = min(vocab - 3, 49405)
    for b in range(batch):
        n = int(torch.randint(1, word_len - 1, (1,), generator=g))       # 1 .. L-2 content tokens
        ids = torch.randint(1, hi, (n,), generator=g)
        word[b, 0] = SOT if vocab > SOT else vocab - 2
        word[b, 1:1 + n] = ids
        word[b, 1 + n] = EOT if vocab > EOT else vocab - 1                # largest id => argmax = EOT slot
    if rect_mask:
        mask = torch.zeros(batch, size, size)
        for b in range(batch):
            y0, x0 = [int(v) for v in torch.randint(0, size // 2, (2,), generator=g)]
            h, w = [int(v) for v in torch.randint(size // 8, size // 2, (2,), generator=g)]
            mask[b, y0:y0 + h, x0:x0 + w] = 1.0
    else:
        mask = (torch.rand(batch, size, size, generator=g) > 0.5).float()
    return img, word, mask.unsqueeze(1)
