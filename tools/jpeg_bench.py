"""Decode rate of the hybrid JPEG path (csrc/jpeg.hip): batches of COCO-sized files (480x640, 4:2:0, quality 85; synthetic
content with photo-like statistics) - host Huffman threads, H2D of the coefficients, the two reconstruction launches - and
Pillow's libjpeg-turbo decoding the same files on one host core beside it.  Prints one JSON line.
    python tools/jpeg_bench.py [--batch 8] [--threads 8] [--iters 50]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cris.pytorch_amd import hip, jpegdec      # noqa: E402
import jpeg_cases                               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    files = [jpeg_cases.encode(jpeg_cases._smooth(rng, 480, 640), quality=85, subsampling=2) for _ in range(args.batch)]
    out = {"batch": args.batch, "threads": args.threads, "file_kB": round(sum(map(len, files)) / len(files) / 1e3, 1), "size": "480x640 4:2:0 q85"}
    t0 = time.time()
    for _ in range(3):
        for f in files:
            jpeg_cases.pil_decode(f)
    out["pillow_1core_ms_per_image"] = round((time.time() - t0) / (3 * len(files)) * 1e3, 3)
    t0 = time.time()
    for _ in range(args.iters):
        jpegdec.decode_coefficients(files, threads=args.threads)
    out["host_huffman_ms_per_batch"] = round((time.time() - t0) / args.iters * 1e3, 3)
    t0 = time.time()
    for _ in range(args.iters):
        jpegdec.decode_coefficients(files, threads=1)
    out["host_huffman_1thread_ms_per_image"] = round((time.time() - t0) / args.iters / len(files) * 1e3, 3)
    if torch.cuda.is_available():
        dev = torch.device("cuda:0")
        for _ in range(3):
            jpegdec.decode_batch(files, dev, threads=args.threads)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(args.iters):
            jpegdec.decode_batch(files, dev, threads=args.threads)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / args.iters
        out["end_to_end_ms_per_batch"] = round(dt * 1e3, 3)
        out["images_per_s"] = round(args.batch / dt, 1)
        # device part alone: events around the reconstruction of resident coefficients
        infos, coef, offs = jpegdec.decode_coefficients(files, threads=args.threads, pin=True)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import ctypes as C
        n = len(files)
        dcoef = coef.to(dev)
        planes = torch.empty(sum(i.plane_bytes for i in infos) + 256 * n, dtype=torch.uint8, device=dev)
        rgb = torch.empty(sum(i.width * i.height * 3 for i in infos) + 256 * n, dtype=torch.uint8, device=dev)
        tab = (hip.JpegImage * n)()
        po = ro = 0
        for i in range(n):
            tab[i].coef, tab[i].planes, tab[i].rgb, tab[i].info = dcoef.data_ptr() + 2 * offs[i], planes.data_ptr() + po, rgb.data_ptr() + ro, infos[i]
            po += (infos[i].plane_bytes + 255) // 256 * 256
            ro += (infos[i].width * infos[i].height * 3 + 255) // 256 * 256
        dtab = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev)
        mb, mp = max(i.total_blocks for i in infos), max(i.width * i.height for i in infos)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            hip.call("cris_jpeg_reconstruct", dtab.data_ptr(), n, mb, mp, s)
        ev0.record()
        for _ in range(args.iters):
            hip.call("cris_jpeg_reconstruct", dtab.data_ptr(), n, mb, mp, s)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / args.iters
        px = sum(i.width * i.height for i in infos)
        alg = sum(2 * i.coef_count + 2 * i.plane_bytes + 3 * i.width * i.height for i in infos)      # coefficients in, planes out + in, RGB out
        out["device_reconstruct_ms_per_batch"] = round(ms, 4)
        out["device_algorithmic_GBps"] = round(alg / ms / 1e6, 1)
        out["device_Mpixel_per_s"] = round(px / ms / 1e3, 1)
    print("JPEGBENCH " + json.dumps(out))


if __name__ == "__main__":
    main()
