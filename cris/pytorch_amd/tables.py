"""Host-side constant tables (numpy): computed once per shape, cached on the device by the engine.

  * sin/cos positional encodings of the decoder (reference model/layers.py:106-152, which rebuilds them
    on the CPU and copies them to the device 12 times per forward);
  * the bicubic resize of the attnpool positional embedding (reference model/clip.py:80-108) written as
    the constant linear map it is: pos_resized[HW] = R[HW, G*G] @ pos[1:], so the device applies R and the
    gradient is R^T (F.interpolate(mode='bicubic', align_corners=False): A = -0.75, border clamped).
"""
import math

import numpy as np


def pos2d_table(C: int, H: int, W: int) -> np.ndarray:
    """[H*W, C] fp32.  First C/2 channels encode the column (w), last C/2 the row (h); sin on even, cos on odd."""
    if C % 4 != 0:
        raise ValueError("Cannot use sin/cos positional encoding with odd dimension (got dim=%d)" % C)
    d = C // 2
    div = np.exp(np.arange(0.0, d, 2, dtype=np.float32) * np.float32(-(math.log(10000.0) / d))).astype(np.float32)
    pw = np.arange(W, dtype=np.float32)[:, None] * div[None, :]          # [W, d/2]
    ph = np.arange(H, dtype=np.float32)[:, None] * div[None, :]          # [H, d/2]
    pe = np.zeros((C, H, W), dtype=np.float32)
    pe[0:d:2] = np.sin(pw).T[:, None, :]
    pe[1:d:2] = np.cos(pw).T[:, None, :]
    pe[d::2] = np.sin(ph).T[:, :, None]
    pe[d + 1::2] = np.cos(ph).T[:, :, None]
    return np.ascontiguousarray(pe.reshape(C, H * W).T)


def pos1d_table(D: int, L: int) -> np.ndarray:
    """[L, D] fp32."""
    if D % 2 != 0:
        raise ValueError("Cannot use sin/cos positional encoding with odd dim (got dim=%d)" % D)
    div = np.exp(np.arange(0, D, 2, dtype=np.float32) * np.float32(-(math.log(10000.0) / D))).astype(np.float32)
    pos = np.arange(L, dtype=np.float32)[:, None]
    pe = np.zeros((L, D), dtype=np.float32)
    pe[:, 0::2] = np.sin(pos * div)
    pe[:, 1::2] = np.cos(pos * div)
    return pe


def _cubic_weights(t: float, A: float = -0.75):
    def c1(x):
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def c2(x):
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A

    return [c2(t + 1), c1(t), c1(1 - t), c2(2 - t)]


def _axis_matrix(n_in: int, n_out: int) -> np.ndarray:
    m = np.zeros((n_out, n_in), dtype=np.float64)
    scale = n_in / n_out
    for o in range(n_out):
        src = scale * (o + 0.5) - 0.5
        i0 = math.floor(src)
        w = _cubic_weights(src - i0)
        for k in range(4):
            idx = min(max(i0 - 1 + k, 0), n_in - 1)
            m[o, idx] += w[k]
    return m


def bicubic_resize_matrix(G: int, H: int, W: int) -> np.ndarray:
    """R [H*W, G*G] with out[oy*W+ox] = sum R[., iy*G+ix] * in[iy, ix]."""
    ry = _axis_matrix(G, H)
    rx = _axis_matrix(G, W)
    return np.einsum("ai,bj->abij", ry, rx).reshape(H * W, G * G).astype(np.float32)
