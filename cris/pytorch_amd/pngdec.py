"""8-bit grayscale PNG masks (reference utils/dataset.py:148-149): `cv2.imdecode(.., IMREAD_GRAYSCALE)` of a
record's `mask` field, decoded by libcris_hip.so's host code (csrc/png.hip: zlib / DEFLATE / row filters).  Host work - the
result is a CPU uint8 tensor that `inputpipe.Preprocessor` uploads together with the image."""
import ctypes as C

import torch

from . import hip


def decode_gray(data: bytes) -> torch.Tensor:
    lib = hip.load()
    buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
    w, h = C.c_int(), C.c_int()
    hip.check(lib.cris_png_gray8_size(buf, len(data), C.byref(w), C.byref(h)), "cris_png_gray8_size")
    out = torch.empty(h.value, w.value, dtype=torch.uint8)
    hip.check(lib.cris_png_decode_gray8(buf, len(data), out.data_ptr(), w.value, h.value), "cris_png_decode_gray8")
    return out
