"""Baseline JPEG decoding for the input pipeline: the bytes of an LMDB record's `img` field in, the RGB
uint8 image in HBM out - `cv2.cvtColor(cv2.imdecode(np.frombuffer(ref['img'], np.uint8), cv2.IMREAD_COLOR), cv2.COLOR_BGR2RGB)`
of the reference's loader (utils/dataset.py:127-129), whose output `inputpipe.preprocess_batch` then warps and normalises
without the pixels ever visiting the host.

Hybrid split (include/cris_hip.h, csrc/jpeg.hip): Huffman decoding into quantised coefficients on host threads (bit-serial),
inverse DCT + chroma upsampling + colour conversion as two launches over the whole ragged batch.  The coefficients cross
PCIe as int16: 3 bytes per pixel at 4:2:0, the same as the decoded image would.  Bit-exact with libjpeg(-turbo) at its
default settings (oracle/jpeg_baseline.py, pinned against Pillow's libjpeg-turbo).  Sequential and progressive Huffman files
are decoded; unsupported ones (arithmetic-coded, lossless, CMYK, other samplings) raise hip.HipLibraryError: there is no CPU
decoder in here.
"""
import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from . import hip
from .hip import ptr


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


def decode_batch(files: Sequence[bytes], device, threads: Optional[int] = None) -> List[torch.Tensor]:
    """JPEG files -> list of uint8 [H, W, 3] RGB tensors on `device` (views of one buffer)"""
    if torch.device(device).type != "cuda":
        raise RuntimeError("jpegdec reconstructs on the GPU only (no CPU fallback)")
    lib = hip.load()
    n = len(files)
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
        out.append(rgb[r_off[i]:r_off[i] + h * w * 3].view(h, w, 3))
    # keep the operands alive until the stream has consumed them
    for t in (dcoef, planes, dtab):
        t.record_stream(torch.cuda.current_stream())
    return out
