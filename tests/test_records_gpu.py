"""Record -> network input on the GPU (cris/pytorch_amd/records.py; reference utils/dataset.py:117-191): records built the way
tools/folder2lmdb.py:50-56 builds them (protocol-5 pickles of {'img': JPEG bytes, 'mask': PNG bytes, 'sents', ...}) go through
RecordPipeline and must equal the reference's per-sample arithmetic assembled from the pinned pieces: Pillow's libjpeg-turbo /
PNG decoders for the files, oracle/input_pipe.py for the warps + normalisation, the committed tokenizer semantics."""
import io
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402

from cris.pytorch_amd import records, tokenizer  # noqa: E402
from oracle import input_pipe as ip  # noqa: E402
import jpeg_cases  # noqa: E402

DEV = torch.device("cuda:0")


def _record(rng, h, w, seg_id, sents, **jpeg_kw):
    img = jpeg_cases._smooth(rng, h, w)
    m = np.zeros((h, w), np.uint8)
    m[h // 4:h // 2 + 3, w // 5:w // 2 + 1] = 255
    b = io.BytesIO()
    Image.fromarray(m, "L").save(b, "PNG")
    rec = {"img": jpeg_cases.encode(img, **jpeg_kw), "mask": b.getvalue(), "cat": 1, "seg_id": seg_id, "img_name": "x.jpg",
           "num_sents": len(sents), "sents": sents}
    return pickle.dumps(rec, protocol=5), m


class _StandInTokenizer:
    """where the CLIP merge list is absent (the GPU box): a deterministic stand-in with the same interface, so that the pixel
    path of the pipeline is still exercised; the real tokenizer has its own bit-exact test (tests/test_tokenizer.py)"""

    def tokenize(self, texts, context_length=77, truncate=False):
        texts = [texts] if isinstance(texts, str) else texts
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [49406] + [1 + (sum(map(ord, w)) % 40000) for w in t.lower().split()][:context_length - 2] + [49407]
            out[i, :len(ids)] = torch.tensor(ids)
        return out


def test_train_and_val_batches_match_the_reference_arithmetic():
    tok = tokenizer.BPETokenizer() if tokenizer.default_merges_path() else _StandInTokenizer()
    rng = np.random.default_rng(0)
    specs = [(120, 160, dict(quality=85, subsampling=2)), (160, 120, dict(quality=90, subsampling=1)),
             (50, 37, dict(quality=75, subsampling=2, progressive=True)), (96, 96, dict(quality=95, subsampling=0))]
    vals, masks = [], []
    for i, (h, w, kw) in enumerate(specs):
        v, m = _record(rng, h, w, 1000 + i, ["the left one", "Woman's umbrella #%d" % i, "zebra closest 2 us"], **kw)
        vals.append(v)
        masks.append(m)
    recs = [records.load_record(v) for v in vals]
    S, L = 96, 17
    pipe = records.RecordPipeline(S, L, DEV, mode="train", tokenizer=tok)
    img, word, mask = pipe(recs, rng=np.random.default_rng(5))
    torch.cuda.synchronize()
    assert img.shape == (4, 3, S, S) and word.shape == (4, L) and mask.shape == (4, S, S) and word.dtype == torch.int64
    choice = np.random.default_rng(5)
    for b, r in enumerate(recs):
        rgb = jpeg_cases.pil_decode(r["img"])
        ref_img, ref_mask, _, _ = ip.preprocess_train(rgb, masks[b], (S, S))
        assert np.array_equal(img[b].cpu().numpy(), ref_img), b
        assert np.array_equal(mask[b].cpu().numpy(), ref_mask), b
        sent = r["sents"][int(choice.integers(r["num_sents"]))]
        assert word[b].tolist() == pipe.tok.tokenize(sent, L, True)[0].tolist()
        assert word[b, 0] == 49406 and int(word[b].argmax()) == int((word[b] == 49407).nonzero()[0])
    # val mode: first sentence, params keyed like the reference's
    vimg, vword, params = records.RecordPipeline(S, L, DEV, mode="val", mask_dir="/m", tokenizer=tok)(recs)
    torch.cuda.synchronize()
    assert torch.equal(vimg, img)
    for b, r in enumerate(recs):
        assert vword[b].tolist() == pipe.tok.tokenize(r["sents"][0], L, True)[0].tolist()
        assert params[b]["mask_dir"] == "/m/%d.png" % r["seg_id"] and params[b]["ori_size"].tolist() == list(masks[b].shape)
        assert params[b]["inverse"].shape == (2, 3)
