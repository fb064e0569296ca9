"""csrc/common.h cris_fast_div(n, d, rd): q = (int)((float)n * rd), then ONE correction step from the remainder - the index
arithmetic of the GEMM prologues and of the pooling / resampling / BatchNorm kernels since round 5 (no 40-instruction integer
division sequences).  Model check in float32 numpy, with the reciprocal perturbed by +-1 ulp (v_rcp_f32's error bound): the
result equals n // d for every dividend below 2^24 as long as the QUOTIENT stays below 2^22 - which holds at every call site
(pixel / tile / vector indices divided by channel-vector counts >= 4, image sizes, tile counts; divisors 1 and 2 are exact)."""
import numpy as np


def fast_div(n, d, rd):
    q = (n.astype(np.float32) * np.float32(rd)).astype(np.int64)
    r = n - q * d
    q = q + (r >= d) - (r < 0)
    return q


def test_fast_div_matches_integer_division():
    rng = np.random.default_rng(0)
    divisors = sorted(set(list(range(1, 600)) + [676, 900, 2704, 10816, 43264, 14400, 3600, 86528, 346112, 5408, 1352, 21632,
                                                 13, 26, 52, 104, 208, 15, 30, 60, 120, 240, 85, 170, 680, 1360, 2711, 65537]))
    for d in divisors:
        hi = min(1 << 24, d << 22)                      # dividends below 2^24 with quotients below 2^22
        n = np.unique(np.concatenate([rng.integers(0, hi, 20000), np.arange(0, min(hi, 4096)),
                                      np.arange(max(0, hi - 4096), hi),
                                      (np.arange(1, 3000) * d).clip(0, hi - 1), (np.arange(1, 3000) * d - 1).clip(0, hi - 1)])).astype(np.int64)
        want = n // d
        rd = np.float32(1.0) / np.float32(d)
        for pert in (rd, np.nextafter(rd, np.float32(0)), np.nextafter(rd, np.float32(2))):
            got = fast_div(n, d, pert)
            bad = np.nonzero(got != want)[0]
            assert bad.size == 0, (d, int(n[bad[0]]), int(got[bad[0]]), int(want[bad[0]]))
