"""Test JPEG files for the decoder tests: synthetic images encoded by Pillow (libjpeg-turbo) over every supported chroma
sampling, several qualities, odd sizes (MCU tails, components of <= 2 columns), optimised Huffman tables and restart intervals.
Deterministic (seeded); the bytes a given Pillow build writes may differ from another build's, which does not matter: every
test decodes the bytes it is given with both sides."""
import io

import numpy as np


def _smooth(rng, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 100 * np.sin(xx / 7.0 + yy / 13.0), 127 + 100 * np.cos(xx / 5.0 - yy / 9.0), (xx * 3 + yy * 2) % 256], -1)
    img = img + rng.normal(0, 8, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def encode(img, **kw):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", **kw)
    return b.getvalue()


def pil_decode(data):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def cases(sizes=((16, 16), (37, 53), (64, 48), (33, 31), (8, 8), (1, 1), (3, 5), (17, 2), (50, 1), (9, 3), (9, 4), (2, 4), (5, 6)),
          qualities=(30, 85, 100), big=None):
    """yields (name, jpeg bytes)"""
    rng = np.random.default_rng(0)
    for (h, w) in sizes:
        for sub in (0, 1, 2):
            for q in qualities:
                for kind in ("smooth", "noise"):
                    img = _smooth(rng, h, w) if kind == "smooth" else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
                    yield "%dx%d_s%d_q%d_%s" % (h, w, sub, q, kind), encode(img, quality=q, subsampling=sub)
    extra = [((40, 56), dict(quality=75, subsampling=2, optimize=True)), ((41, 57), dict(quality=90, subsampling=2, restart_marker_blocks=3)),
             ((41, 57), dict(quality=90, subsampling=1, restart_marker_rows=1)),
             ((24, 40), dict(quality=95, subsampling=2, optimize=True, restart_marker_blocks=1))]
    for (h, w), kw in extra:
        yield "%dx%d_%s" % (h, w, "_".join("%s%s" % (k[:4], v) for k, v in sorted(kw.items()))), encode(_smooth(rng, h, w), **kw)
    for (h, w) in ((20, 33), (8, 8)):
        yield "%dx%d_gray" % (h, w), encode(_smooth(rng, h, w)[..., 0], quality=80)
    if big is not None:
        h, w = big
        yield "%dx%d_s2_q85_photo" % (h, w), encode(_smooth(rng, h, w), quality=85, subsampling=2)
