"""PNG mask decoding (csrc/png.hip, host only): the library, the oracle (oracle/png_gray.py: zlib + numpy) and Pillow's decoder
agree byte for byte on 8-bit gray files of every zlib level (stored / fixed / dynamic blocks), both encoder filter strategies,
mask-like and noisy content; other PNG flavours and damaged files are refused."""
import io

import numpy as np
import pytest

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402

from cris.pytorch_amd import hip, pngdec  # noqa: E402
from oracle import png_gray  # noqa: E402


def _mask(rng, h, w):
    m = np.zeros((h, w), np.uint8)
    for _ in range(3):
        y0, x0 = rng.integers(0, h), rng.integers(0, w)
        m[y0:y0 + rng.integers(1, h + 1), x0:x0 + rng.integers(1, w + 1)] = 255
    return m


def _files():
    rng = np.random.default_rng(0)
    for (h, w) in ((1, 1), (7, 5), (64, 48), (120, 160), (480, 640), (3, 700)):
        for kind in ("mask", "noise", "ramp"):
            img = _mask(rng, h, w) if kind == "mask" else (rng.integers(0, 256, (h, w), dtype=np.uint8) if kind == "noise"
                                                            else ((np.arange(w)[None, :] * 3 + np.arange(h)[:, None] * 5) % 256).astype(np.uint8))
            for kw in (dict(compress_level=0), dict(compress_level=1), dict(compress_level=6), dict(compress_level=9, optimize=True)):
                b = io.BytesIO()
                Image.fromarray(img, "L").save(b, "PNG", **kw)
                yield "%dx%d_%s_%s" % (h, w, kind, kw), b.getvalue(), img


def test_library_oracle_and_pillow_agree():
    n = 0
    for name, data, img in _files():
        got = pngdec.decode_gray(data).numpy()
        assert np.array_equal(got, img), name                                    # lossless: the encoder's input
        assert np.array_equal(got, np.asarray(Image.open(io.BytesIO(data)))), name
        if img.size <= 64 * 48:
            assert np.array_equal(png_gray.decode_gray(data), img), name
        n += 1
    assert n == 72


def test_other_flavours_and_damage_are_refused():
    rng = np.random.default_rng(1)
    b = io.BytesIO()
    Image.fromarray(rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)).save(b, "PNG")
    with pytest.raises(hip.HipLibraryError, match="8-bit grayscale"):
        pngdec.decode_gray(b.getvalue())
    with pytest.raises(hip.HipLibraryError, match="not a PNG"):
        pngdec.decode_gray(b"\xff\xd8\xff\xe0" + bytes(64))
    b = io.BytesIO()
    Image.fromarray(_mask(rng, 40, 40), "L").save(b, "PNG")
    good = bytearray(b.getvalue())
    refused = 0
    for it in range(400):
        bad = bytearray(good)
        for _ in range(rng.integers(1, 4)):
            bad[rng.integers(33, len(bad) - 12)] = rng.integers(0, 256)
        try:
            out = pngdec.decode_gray(bytes(bad))
            assert out.shape == (40, 40)
        except hip.HipLibraryError:
            refused += 1
    assert refused > 200                                                         # Adler-32 / DEFLATE structure catch damage
    with pytest.raises(hip.HipLibraryError):
        pngdec.decode_gray(bytes(good[:60]))


def test_property_round_trip():
    """hypothesis: any 8-bit gray image, any zlib level -> the encoder's input comes back"""
    pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    from hypothesis.extra import numpy as hnp

    @settings(max_examples=150, deadline=None, derandomize=True)
    @given(hnp.arrays(np.uint8, st.tuples(st.integers(1, 40), st.integers(1, 70))), st.integers(0, 9))
    def check(img, level):
        b = io.BytesIO()
        Image.fromarray(img, "L").save(b, "PNG", compress_level=level)
        assert np.array_equal(pngdec.decode_gray(b.getvalue()).numpy(), img)

    check()
