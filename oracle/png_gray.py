"""TEST INFRASTRUCTURE - 8-bit grayscale PNG decoding restated with the standard library (zlib) + numpy row filters, for the
mask files of the reference's records (utils/dataset.py:148-149, tools/data_process.py:115-117).  Lossless: pinned against
Pillow's decoder in tests/test_png.py.  Only tests/ may import this."""
import struct
import zlib

import numpy as np


def decode_gray(data: bytes) -> np.ndarray:
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            w, h, depth, ctype, _, _, lace = struct.unpack(">IIBBBBB", body)
            assert (depth, ctype, lace) == (8, 0, 0)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, w + 1)
    out = np.zeros((h, w), dtype=np.uint8)
    for y in range(h):
        f, src = int(raw[y, 0]), raw[y, 1:].astype(np.int64)
        up = out[y - 1].astype(np.int64) if y else np.zeros(w, dtype=np.int64)
        if f == 0:
            row = src
        elif f == 2:
            row = src + up
        else:
            row = np.zeros(w, dtype=np.int64)
            for x in range(w):
                a = row[x - 1] if x else 0
                b, c = up[x], (up[x - 1] if x else 0)
                if f == 1:
                    pr = a
                elif f == 3:
                    pr = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                row[x] = (src[x] + pr) & 255
        out[y] = row & 255
    return out
