"""LMDB record -> network input, for a whole batch, on this package's own path: what
`RefDataset.__getitem__` + the default collate do per sample in 32 loader workers (reference utils/dataset.py:117-191), as one
call per batch whose pixel work runs on the GPU:

    record bytes --pickle--> {'img': JPEG bytes, 'mask': PNG bytes, 'sents': [...], 'num_sents', 'seg_id', ...}
      img   --jpegdec (host Huffman threads + 2 launches)--> uint8 RGB in HBM --inputpipe (1 launch)--> float32 [B, 3, S, S]
      mask  --pngdec (host)--> uint8 --inputpipe (same launch)--> float32 [B, S, S]                              (train mode)
      sent  --tokenizer--> int64 [B, L]

`load_record` reads one value of the LMDB the reference's tools/folder2lmdb.py:50-64 writes (`pickle5.dumps(obj, protocol=5)`;
the reference's own reader still calls the long-removed `pyarrow.deserialize`, utils/dataset.py:87-92 - protocol-5 pickles are
read by the standard library since Python 3.8).  `LmdbRecords` opens the LMDB environment itself (reference `RefDataset._init_db` /
`__getitem__` up to the unpickling, utils/dataset.py:112-133) through this package's own read-only container reader
(`lmdbfile.LmdbReader`: the `lmdb` C extension is not available on the training image).  Modes follow the reference: 'train' (random sentence, image + mask + tokens),
'val' (first sentence, image + tokens + params), 'test' (image + params with every sentence).
"""
import pickle
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import inputpipe, jpegdec, pngdec
from .tokenizer import BPETokenizer


def load_record(value: bytes) -> dict:
    """one LMDB value -> the record dict (tools/folder2lmdb.py:50-56).
    TRUST: the reference's records are pickles, and unpickling executes whatever the file says - exactly as the reference's own
    loader does (utils/dataset.py:92).  Only feed this datasets you built yourself (tools/folder2lmdb.py); the JPEG / PNG
    parsers behind it are hardened against corrupt input, the pickle layer cannot be."""
    rec = pickle.loads(value)
    if not isinstance(rec, dict) or "img" not in rec:
        raise ValueError("not a CRIS record (expected a dict with 'img', 'mask', 'sents', ...)")
    return rec


class LmdbRecords:
    """The records of one dataset split, indexable like the reference's `RefDataset` before its image processing:
    `len(ds)` = the pickled `__len__` entry, `ds[i]` = the record dict stored under `__keys__[i]` (utils/dataset.py:112-133,
    written by tools/folder2lmdb.py:36-68).  `path` is what the reference passes to `lmdb.open` (a directory holding `data.mdb`,
    or the file).  Opened lazily, like the reference, so that an instance can be handed to worker processes."""

    def __init__(self, path: str):
        self.path = path
        self._db = None
        self.length = None
        self.keys = None

    def _init_db(self):
        from .lmdbfile import LmdbReader
        self._db = LmdbReader(self.path)
        n, keys = self._db.get(b"__len__"), self._db.get(b"__keys__")
        if n is None or keys is None:
            raise ValueError("%s: no __len__ / __keys__ entries (not written by tools/folder2lmdb.py?)" % self.path)
        self.length, self.keys = pickle.loads(n), pickle.loads(keys)

    def __len__(self):
        if self._db is None:
            self._init_db()
        return self.length

    def raw(self, index: int) -> bytes:
        """the stored value of record `index` (what `txn.get(self.keys[index])` returns)"""
        if self._db is None:
            self._init_db()
        value = self._db.get(self.keys[index])
        if value is None:
            raise KeyError("record %d (key %r) is listed in __keys__ but missing" % (index, self.keys[index]))
        return value

    def __getitem__(self, index: int) -> dict:
        return load_record(self.raw(index))

    def batch(self, indices: Sequence[int]) -> List[dict]:
        """records for RecordPipeline.__call__"""
        return [self[int(i)] for i in indices]

    def close(self):
        if self._db is not None:
            self._db.close()
            self._db = None

    def __getstate__(self):                      # (the mmap does not travel to worker processes: reopen there)
        return {"path": self.path}

    def __setstate__(self, st):
        self.__init__(st["path"])


class RecordPipeline:
    def __init__(self, input_size: int, word_length: int, device, mode: str = "train", tokenizer: Optional[BPETokenizer] = None,
                 mask_dir: str = "", decode_threads: Optional[int] = None):
        assert mode in ("train", "val", "test")
        self.mode, self.word_length, self.device, self.mask_dir = mode, word_length, torch.device(device), mask_dir
        self.pre = inputpipe.Preprocessor((input_size, input_size), device)
        self.tok = tokenizer if tokenizer is not None else BPETokenizer()
        self.threads = decode_threads

    def __call__(self, records: Sequence[dict], rng: Optional[np.random.Generator] = None):
        """train: (img [B,3,S,S] f32 cuda, word [B,L] i64 cuda, mask [B,S,S] f32 cuda)  - what engine.train's loop body gets after
        its .cuda() calls (engine/engine.py:39-42 adds the mask's channel dimension itself);
        val:   (img, word, params) ; test: (img, params) with params as lists of per-sample dicts keyed like the reference's."""
        rng = rng if rng is not None else np.random.default_rng()
        images = jpegdec.decode_batch([r["img"] for r in records], self.device, threads=self.threads)
        masks = [pngdec.decode_gray(r["mask"]) for r in records] if self.mode == "train" else None
        img, mask, _, invs = self.pre(images, masks)
        if self.mode == "test":
            return img, [self._params(r, im, inv, test=True) for r, im, inv in zip(records, images, invs)]
        if self.mode == "train":
            sents = [r["sents"][int(rng.integers(r["num_sents"]))] for r in records]      # np.random.choice(num_sents)
        else:
            sents = [r["sents"][0] for r in records]
        word = self.tok.tokenize(sents, self.word_length, True).to(self.device, non_blocking=True)
        if self.mode == "train":
            return img, word, mask
        return img, word, [self._params(r, im, inv, test=False) for r, im, inv in zip(records, images, invs)]

    def _params(self, rec, image, inv, test):
        import os
        p = {"mask_dir": os.path.join(self.mask_dir, str(rec["seg_id"]) + ".png"), "inverse": inv,
             "ori_size": np.array([int(image.shape[0]), int(image.shape[1])])}
        if test:
            p.update(ori_img=image, seg_id=rec["seg_id"], sents=rec["sents"])       # ori_img: RGB uint8 on the GPU (the reference keeps BGR on the host)
        return p
