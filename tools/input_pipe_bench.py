"""throughput of the GPU input preprocessing (cris/pytorch_amd/inputpipe.py): batches of 8 decoded 480x640 RGB images + masks
-> [8, 3, 416, 416] float32 + [8, 416, 416] masks.  Prints samples/s with the uint8 arrays already on the device and with
the host->device copies included (pinned host memory).   python tools/input_pipe_bench.py [iters]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd.inputpipe import Preprocessor      # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(0)
imgs_h = [torch.from_numpy(rng.integers(0, 255, (480, 640, 3), dtype=np.uint8)).pin_memory() for _ in range(8)]
masks_h = [torch.from_numpy((rng.random((480, 640)) > 0.5).astype(np.uint8) * 255).pin_memory() for _ in range(8)]
imgs_d, masks_d = [t.cuda() for t in imgs_h], [t.cuda() for t in masks_h]
pre = Preprocessor((416, 416))
for name, (a, b) in (("device-resident uint8", (imgs_d, masks_d)), ("pinned host uint8 (incl. H2D)", (imgs_h, masks_h))):
    for _ in range(5):
        pre(a, b)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        pre(a, b)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / iters
    out_bytes = 8 * (3 * 416 * 416 * 4 + 416 * 416 * 4)
    in_bytes = 8 * (480 * 640 * 4)
    print("%-32s %.3f ms / batch of 8 = %.0f samples/s (%.1f GB/s of input bytes + output bytes)" % (
        name, dt * 1e3, 8 / dt, (out_bytes + in_bytes) / dt / 1e9))
