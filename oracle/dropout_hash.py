"""TEST INFRASTRUCTURE ONLY - counter-based dropout mask, numpy restatement.

The reference draws dropout masks from torch's Philox stream (nn.Dropout / MHA dropout,
reference model/layers.py:202-219) which no other implementation can reproduce.  The HIP path
instead derives every keep/drop decision from (seed, stream, element index) with the integer
hash below (csrc/common.h: cris_keep()).  This file restates that hash with numpy uint32
arithmetic so the oracle can apply *the same* masks and parity holds bit-for-bit on the mask
(an index op) and to floating-point tolerance on everything downstream.

    h    = seed ^ (stream * 0x9E3779B9)
    v    = idx * 0x9E3779B1 + h                      (uint32 wraparound)
    v    = fmix32(v)                                 (murmur3 finaliser)
    keep = v >= floor(p * 2**32)
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _mul32(a, b):
    return ((a.astype(np.uint64) * np.uint64(b)) & _M32).astype(np.uint32)


def fmix32(v):
    v = v.astype(np.uint32)
    v ^= v >> np.uint32(16)
    v = _mul32(v, 0x85EBCA6B)
    v ^= v >> np.uint32(13)
    v = _mul32(v, 0xC2B2AE35)
    v ^= v >> np.uint32(16)
    return v


def threshold(p: float) -> int:
    return min(int(p * 4294967296.0), 0xFFFFFFFF)


def keep_mask(seed: int, stream: int, n: int, p: float) -> np.ndarray:
    """bool[n]: element idx in [0, n) survives dropout with probability 1-p."""
    idx = np.arange(n, dtype=np.uint64)
    h = np.uint32((seed ^ ((stream * 0x9E3779B9) & 0xFFFFFFFF)) & 0xFFFFFFFF)
    v = ((idx * np.uint64(0x9E3779B1) + np.uint64(h)) & _M32).astype(np.uint32)
    return fmix32(v) >= np.uint32(threshold(p))


def keep_mask_torch(seed: int, stream: int, n: int, p: float, device):
    """keep_mask() with torch int64 arithmetic on `device` (bit-identical; tests/test_oracle_device.py): lets the oracle run on
    the GPU, where 30 M decisions per attention site and step through numpy would dominate a 100-step trajectory"""
    import torch
    M = 0xFFFFFFFF
    h = (seed ^ ((stream * 0x9E3779B9) & M)) & M
    v = (torch.arange(n, dtype=torch.int64, device=device) * 0x9E3779B1 + h) & M
    v = v ^ (v >> 16)
    v = (v * 0x85EBCA6B) & M
    v = v ^ (v >> 13)
    v = (v * 0xC2B2AE35) & M
    v = v ^ (v >> 16)
    return v >= threshold(p)
