"""Baseline JPEG decoding for the input pipeline: the bytes of an LMDB record's `img` field in, the RGB
uint8 image in HBM out - `cv2.cvtColor(cv2.imdecode(np.frombuffer(ref['img'], np.uint8), cv2.IMREAD_COLOR), cv2.COLOR_BGR2RGB)`
of the reference's loader (utils/dataset.py:127-129), whose output `inputpipe.preprocess_batch` then warps and normalises
without the pixels ever visiting the host.

Hybrid split (include/cris_hip.h, csrc/jpeg.hip): Huffman decoding into quantised coefficients on host threads (bit-serial),
inverse DCT + chroma upsampling + colour conversion as two launches over the whole ragged batch.  The coefficients cross
PCIe as int16: 3 bytes per pixel at 4:2:0, the same as the decoded image would.  Bit-exact with libjpeg(-turbo) at its
default settings (oracle/jpeg_baseline.py, pinned against Pillow's libjpeg-turbo); an EXIF orientation tag is honoured the way
cv2.imdecode honours it (the turned array is returned).  Sequential and progressive Huffman files
are decoded; unsupported ones (arithmetic-coded, lossless, CMYK, other samplings) raise hip.HipLibraryError: there is no CPU
decoder in here.
"""
import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from . import hip
from .hip import ptr


def exif_orientation(data: bytes) -> int:
    """EXIF orientation tag (0x0112) of a JPEG file, 1 (= upright) when there is none.  cv2.imdecode applies it for
    IMREAD_COLOR (OpenCV >= 3.1; `IMREAD_IGNORE_ORIENTATION` is not among the reference's flags, utils/dataset.py:127-128), so
    the decoded array the loader sees is already turned."""
    pos, n = 2, len(data)
    while pos + 4 <= n and data[pos] == 0xFF:
        m = data[pos + 1]
        if m == 0xDA or m == 0xD9:
            break
        if m == 0xFF:                                   # fill byte
            pos += 1
            continue
        seg_len = (data[pos + 2] << 8) | data[pos + 3]
        if m == 0xE1 and data[pos + 4:pos + 10] == b"Exif\x00\x00":
            t = data[pos + 10:pos + 2 + seg_len]
            if len(t) < 14 or t[:2] not in (b"II", b"MM"):
                return 1
            big = t[:2] == b"MM"
            u16 = lambda o: int.from_bytes(t[o:o + 2], "big" if big else "little")      # noqa: E731
            u32 = lambda o: int.from_bytes(t[o:o + 4], "big" if big else "little")      # noqa: E731
            if u16(2) != 42:
                return 1
            ifd = u32(4)
            if ifd + 2 > len(t):
                return 1
            for k in range(u16(ifd)):
                e = ifd + 2 + 12 * k
                if e + 12 > len(t):
                    break
                if u16(e) == 0x0112:
                    v = u16(e + 8)
                    return v if 1 <= v <= 8 else 1
            return 1
        pos += 2 + seg_len
    return 1


def apply_orientation(img: torch.Tensor, orientation: int) -> torch.Tensor:
    """[H, W, C] turned the way OpenCV's ExifTransform / PIL's exif_transpose turn it (pure data movement)"""
    if orientation == 2:
        return img.flip(1)
    if orientation == 3:
        return img.flip(0, 1)
    if orientation == 4:
        return img.flip(0)
    if orientation == 5:
        return img.transpose(0, 1)
    if orientation == 6:
        return img.transpose(0, 1).flip(1)
    if orientation == 7:
        return img.transpose(0, 1).flip(0, 1)
    if orientation == 8:
        return img.transpose(0, 1).flip(0)
    return img


def read_header(data: bytes) -> hip.JpegInfo:
    """geometry of one file without decoding it (host only)"""
    info = hip.JpegInfo()
    buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
    hip.check(hip.load().cris_jpeg_read_header(buf, len(data), C.byref(info)), "cris_jpeg_read_header")
    return info


def decode_coefficients(files: Sequence[bytes], infos=None, threads: Optional[int] = None, pin: bool = False):
    """Host half: (infos, coefficient tensor int16 [sum coef_count] on the CPU, offsets).  One file per host thread."""
    lib = hip.load()
    n = len(files)
    bufs = [(C.c_ubyte * len(f)).from_buffer_copy(f) for f in files]
    if infos is None:
        infos = (hip.JpegInfo * n)()
        for i in range(n):
            hip.check(lib.cris_jpeg_read_header(bufs[i], len(files[i]), C.byref(infos[i])), "cris_jpeg_read_header (image %d)" % i)
    offs, total = [], 0
    for i in range(n):
        offs.append(total)
        total += (infos[i].coef_count + 127) // 128 * 128            # 256-byte aligned starts
    coef = torch.empty(total, dtype=torch.int16, pin_memory=pin)
    base = coef.data_ptr()
    datas = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    sizes = (C.c_size_t * n)(*[len(f) for f in files])
    outs = (C.c_void_p * n)(*[base + 2 * o for o in offs])
    if threads is None:
        threads = min(n, os.cpu_count() or 1)
    hip.check(lib.cris_jpeg_decode_coefficients_batch(n, datas, sizes, infos, outs, int(threads)), "cris_jpeg_decode_coefficients_batch")
    return infos, coef, offs


def decode_batch(files: Sequence[bytes], device, threads: Optional[int] = None, apply_exif: bool = True) -> List[torch.Tensor]:
    """JPEG files -> list of uint8 [H, W, 3] RGB tensors on `device` (views of one buffer; files that carry an EXIF orientation
    other than 1 come back turned, as cv2.imdecode returns them - apply_exif=False = IMREAD_IGNORE_ORIENTATION)"""
    if torch.device(device).type != "cuda":
        raise RuntimeError("jpegdec reconstructs on the GPU only (no CPU fallback)")
    lib = hip.load()
    n = len(files)
    if n == 0:
        return []
    infos, coef, offs = decode_coefficients(files, threads=threads, pin=True)
    dcoef = coef.to(device, non_blocking=True)
    p_off, r_off, ptot, rtot = [], [], 0, 0
    for i in range(n):
        p_off.append(ptot)
        r_off.append(rtot)
        ptot += (infos[i].plane_bytes + 255) // 256 * 256
        rtot += (infos[i].width * infos[i].height * 3 + 255) // 256 * 256
    planes = torch.empty(ptot, dtype=torch.uint8, device=device)
    rgb = torch.empty(rtot, dtype=torch.uint8, device=device)
    tab = (hip.JpegImage * n)()
    for i in range(n):
        tab[i].coef = dcoef.data_ptr() + 2 * offs[i]
        tab[i].planes = planes.data_ptr() + p_off[i]
        tab[i].rgb = rgb.data_ptr() + r_off[i]
        tab[i].info = infos[i]
    dtab = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(device)
    max_blocks = max(infos[i].total_blocks for i in range(n))
    max_pixels = max(infos[i].width * infos[i].height for i in range(n))
    hip.call("cris_jpeg_reconstruct", ptr(dtab), n, max_blocks, max_pixels, torch.cuda.current_stream().cuda_stream)
    out = []
    for i in range(n):
        h, w = infos[i].height, infos[i].width
        t = rgb[r_off[i]:r_off[i] + h * w * 3].view(h, w, 3)
        o = exif_orientation(files[i]) if apply_exif else 1
        out.append(t if o == 1 else apply_orientation(t, o).contiguous())
    # keep the operands alive until the stream has consumed them
    for t in (dcoef, planes, dtab):
        t.record_stream(torch.cuda.current_stream())
    return out
