"""JPEG decoding on the GPU (reference utils/dataset.py:127-129): the whole ragged case list decoded by ONE
cris_jpeg_reconstruct call equals Pillow's libjpeg-turbo (the library the oracle is pinned to, tests/test_jpeg_host.py) and
the oracle bit for bit; decode -> letter-box warp -> normalise chained on the device equals the same chain fed with
Pillow-decoded arrays."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PIL = pytest.importorskip("PIL")

from cris.pytorch_amd import inputpipe, jpegdec  # noqa: E402
from oracle import jpeg_baseline as J  # noqa: E402
import jpeg_cases  # noqa: E402

DEV = torch.device("cuda:0")


def test_batch_decode_is_bit_exact_with_libjpeg_turbo():
    files = list(jpeg_cases.cases(big=(480, 640)))
    out = jpegdec.decode_batch([d for _, d in files], DEV, threads=4)
    torch.cuda.synchronize()
    assert len(out) == len(files)
    for (name, data), t in zip(files, out):
        ref = jpeg_cases.pil_decode(data)
        got = t.cpu().numpy()
        assert got.shape == ref.shape and np.array_equal(got, ref), name
    for (name, data), t in list(zip(files, out))[::17]:                      # the restatement itself (pure Python: a sample)
        if t.shape[0] * t.shape[1] <= 64 * 64:
            assert np.array_equal(t.cpu().numpy(), J.decode(data)), name


def test_decode_feeds_the_preprocessing_kernel():
    rng = np.random.default_rng(3)
    files = [d for n, d in jpeg_cases.cases(sizes=((120, 160), (160, 120), (50, 37)), qualities=(85,)) if "smooth" in n]
    files.append(jpeg_cases.encode(rng.integers(0, 256, (96, 96), dtype=np.uint8), quality=90))           # a gray file
    pre = inputpipe.Preprocessor((96, 96), DEV)
    a, _, mats, _ = pre(jpegdec.decode_batch(files, DEV))
    a = a.clone()
    b, _, mats_b, _ = pre([jpeg_cases.pil_decode(d) for d in files])
    torch.cuda.synchronize()
    assert torch.equal(a, b) and all(np.array_equal(x, y) for x, y in zip(mats, mats_b))
