"""Test JPEG files for the decoder tests: synthetic images encoded by Pillow (libjpeg-turbo) over every supported chroma
sampling, several qualities, odd sizes (MCU tails, components of <= 2 columns), optimised Huffman tables, restart intervals,
sequential and progressive.
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
    # progressive files (SOF2): DC / AC first passes and refinements, non-interleaved AC scans over the components' real blocks
    for (h, w) in ((37, 53), (33, 31), (8, 8), (1, 1), (17, 2), (9, 4), (64, 80)):
        for sub in (0, 1, 2):
            for q in (30, 90, 100):
                img = _smooth(rng, h, w) if (h + sub + q) % 2 else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
                yield "%dx%d_s%d_q%d_prog" % (h, w, sub, q), encode(img, quality=q, subsampling=sub, progressive=True)
    yield "40x56_prog_opt", encode(_smooth(rng, 40, 56), quality=75, subsampling=2, progressive=True, optimize=True)
    yield "41x57_prog_rst", encode(_smooth(rng, 41, 57), quality=90, subsampling=2, progressive=True, restart_marker_blocks=2)
    yield "20x33_gray_prog", encode(_smooth(rng, 20, 33)[..., 0], quality=80, progressive=True)
    # sequential files with one scan per component (written by to_non_interleaved below)
    for (h, w, sub, q) in ((37, 53, 2, 85), (33, 31, 1, 50), (16, 16, 0, 95), (17, 2, 2, 90), (24, 40, 2, 100)):
        yield "%dx%d_s%d_q%d_nonint" % (h, w, sub, q), to_non_interleaved(encode(_smooth(rng, h, w), quality=q, subsampling=sub))
    if big is not None:
        h, w = big
        yield "%dx%d_s2_q85_photo" % (h, w), encode(_smooth(rng, h, w), quality=85, subsampling=2)
        yield "%dx%d_s2_q85_photo_prog" % (h, w), encode(_smooth(rng, h, w), quality=85, subsampling=2, progressive=True)


def to_non_interleaved(data):
    """Re-writes a sequential file (standard Huffman tables, no restart interval) with ONE SCAN PER COMPONENT - a legal layout
    (T.81 A.2: non-interleaved scans cover each component's own blocks, ceil(width_i / 8) x ceil(height_i / 8), in raster
    order) that Pillow's encoder never produces.  A small Huffman encoder over the coefficients the oracle decodes."""
    from oracle import jpeg_baseline as J
    h = J.parse(data)
    assert not h.multiscan and not h.restart_interval
    coefs = J.entropy_decode(data, h)
    ns = len(h.comps)
    head = data[:h.scan_start - (2 + 6 + 2 * ns)]
    inv = np.argsort(J.ZIGZAG)                     # natural position -> zigzag position

    def enc_table(tab):
        return {sym: (ln, code) for (ln, code), sym in tab.items()}

    out = bytearray(head)
    for ci, c in enumerate(h.comps):
        dc, ac = enc_table(h.dc[c["td"]]), enc_table(h.ac[c["ta"]])
        out += bytes([0xFF, 0xDA, 0, 8, 1, c["id"], (c["td"] << 4) | c["ta"], 0, 63, 0])
        acc, nb, pred = 0, 0, 0
        body = bytearray()

        def put(code, ln):
            nonlocal acc, nb
            acc = (acc << ln) | (code & ((1 << ln) - 1))
            nb += ln
            while nb >= 8:
                b = (acc >> (nb - 8)) & 0xFF
                body.append(b)
                if b == 0xFF:
                    body.append(0)
                nb -= 8

        def put_value(v):
            s = int(abs(v)).bit_length()
            return s, (v if v >= 0 else v + (1 << s) - 1)

        for by in range(-(-c["dh"] // 8)):
            for bx in range(-(-c["dw"] // 8)):
                blk = coefs[ci][by, bx].astype(int)
                zz = np.zeros(64, dtype=int)
                zz[inv] = blk                       # zz[zigzag position] = coefficient
                d = int(zz[0]) - pred
                pred = int(zz[0])
                s, bits = put_value(d)
                put(dc[s][1], dc[s][0])
                if s:
                    put(bits, s)
                run = 0
                last = max([k for k in range(1, 64) if zz[k] != 0], default=0)
                for k in range(1, last + 1):
                    if zz[k] == 0:
                        run += 1
                        continue
                    while run > 15:
                        put(ac[0xF0][1], ac[0xF0][0])
                        run -= 16
                    s, bits = put_value(int(zz[k]))
                    put(ac[(run << 4) | s][1], ac[(run << 4) | s][0])
                    put(bits, s)
                    run = 0
                if last < 63:
                    put(ac[0][1], ac[0][0])
        if nb:
            put((1 << (8 - nb)) - 1, 8 - nb)        # pad the last byte with ones
        out += body
    out += b"\xff\xd9"
    return bytes(out)
