"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's evaluation post-processing
(reference engine/engine.py:100-117 `validate`, :171-188 `inference`):

    pred = sigmoid(model(img, text))                                              engine.py:100-101
    pred = F.interpolate(pred, size=img.shape[-2:], mode='bicubic', align_corners=True)      :102-106
    pred = cv2.warpAffine(pred, mat, (w, h), flags=cv2.INTER_CUBIC, borderValue=0.)          :114-116
    pred = pred > 0.35 ; iou = sum(pred & mask) / (sum(pred | mask) + 1e-6)                  :117-123

PINNING.  Steps 1, 2 and 4 are torch / numpy arithmetic: `upsample_bicubic` below is checked against torch's own
F.interpolate (tests/test_eval_post.py), the IoU is integer counting.  Step 3 lives in OpenCV (cv2 4.x `warpAffine`,
modules/imgproc/src/imgwarp.cpp), a third-party dependency that is NOT in /root/reference (requirement.txt:
`opencv-python`, unpinned) and NOT installed in this image: `warp_affine_cubic` restates its published algorithm -
the matrix is inverted in double (no WARP_INVERSE_MAP flag), source coordinates are computed in fixed point (AB_BITS 10)
and quantised to 1/32 pixel (INTER_BITS 5), the 4x4 bicubic weights (A = -0.75) come from a 32 x 32 table of float
products, taps outside the image contribute borderValue 0 - and is **parity unpinned** against cv2 itself.  Its anchors
are properties any correct warp has (identity, integer translations, the dataset's own forward/inverse matrix pair
`utils/dataset.py:190-205`) and - round 6 - an INDEPENDENT implementation of the same sampling rule: torch's
F.grid_sample(bicubic, zeros, align_corners=True) (Keys kernel A = -0.75, integer pixel coordinates, zero border) on exact
coordinates, which the restatement matches to OpenCV's own 1/32-pixel coordinate quantisation under letter-box, rotation and
anisotropic matrices (tests/test_eval_post.py).  That pins the algorithm class, not cv2's bits.
"""
import numpy as np

AB_BITS = 10
AB_SCALE = 1 << AB_BITS
INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x.astype(np.float32)))


def _cubic_coeffs_torch(t):
    """torch's upsample_bicubic2d coefficients (A = -0.75) for fractional offset t in [0,1): weights of taps -1, 0, +1, +2"""
    A = -0.75

    def c1(x):            # |x| <= 1
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def c2(x):            # 1 < |x| < 2
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A

    return np.stack([c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)], -1)


def upsample_bicubic(x, H, W):
    """F.interpolate(x[None, None], size=(H, W), mode='bicubic', align_corners=True) for one [h, w] float32 map: source
    coordinate = dst * (in - 1) / (out - 1), taps at floor-1 .. floor+2 with indices clamped to the image."""
    x = x.astype(np.float32)
    h, w = x.shape

    def axis(n_in, n_out):
        scale = np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(0)
        src = np.arange(n_out, dtype=np.float32) * scale
        i0 = np.floor(src).astype(np.int64)
        t = (src - i0.astype(np.float32)).astype(np.float32)
        idx = np.clip(i0[:, None] + np.arange(-1, 3)[None, :], 0, n_in - 1)
        return idx, _cubic_coeffs_torch(t).astype(np.float32)

    iy, cy = axis(h, H)
    ix, cx = axis(w, W)
    # torch interpolates along x inside each of the 4 rows, then along y
    rows = x[iy]                                            # [H, 4, w]
    tx = (rows[:, :, ix] * cx[None, None, :, :]).astype(np.float32)       # [H, 4, W, 4]
    rx = ((tx[..., 0] + tx[..., 1]) + tx[..., 2]) + tx[..., 3]
    ty = (rx * cy[:, :, None]).astype(np.float32)
    return (((ty[:, 0] + ty[:, 1]) + ty[:, 2]) + ty[:, 3]).astype(np.float32)


def invert_affine(mat):
    """cv::invertAffineTransform in double (imgwarp.cpp): the map the warp applies per destination pixel"""
    m = np.asarray(mat, np.float64).reshape(2, 3)
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22, A12, A21 = m[1, 1] * D, m[0, 0] * D, -m[0, 1] * D, -m[1, 0] * D
    b1 = -A11 * m[0, 2] - A12 * m[1, 2]
    b2 = -A21 * m[0, 2] - A22 * m[1, 2]
    return np.array([[A11, A12, b1], [A21, A22, b2]], np.float64)


def cubic_table():
    """[32][4] float32 coefficients of cv::interpolateCubic (A = -0.75) for the 32 sub-pixel positions"""
    A = np.float32(-0.75)
    t = (np.arange(INTER_TAB_SIZE, dtype=np.float32) / np.float32(INTER_TAB_SIZE)).astype(np.float32)
    c0 = ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A
    c1 = ((A + 2) * t - (A + 3)) * t * t + 1
    c2 = ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1
    c3 = np.float32(1.0) - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], -1).astype(np.float32)


def warp_coords(mat, w_out, h_out):
    """integer source position (of tap 0, i.e. one left / above the base pixel) and the two 5-bit fractions per destination
    pixel, exactly as cv::warpAffine computes them: fixed point with AB_BITS, rounding term AB_SCALE / INTER_TAB_SIZE / 2"""
    M = invert_affine(mat)
    x = np.arange(w_out, dtype=np.float64)
    y = np.arange(h_out, dtype=np.float64)
    rnd = lambda v: np.rint(v).astype(np.int64)             # cvRound: to nearest, ties to even
    adelta, bdelta = rnd(M[0, 0] * x * AB_SCALE), rnd(M[1, 0] * x * AB_SCALE)
    round_delta = AB_SCALE // INTER_TAB_SIZE // 2
    X0 = rnd((M[0, 1] * y + M[0, 2]) * AB_SCALE) + round_delta
    Y0 = rnd((M[1, 1] * y + M[1, 2]) * AB_SCALE) + round_delta
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    return (X >> INTER_BITS) - 1, (Y >> INTER_BITS) - 1, X & (INTER_TAB_SIZE - 1), Y & (INTER_TAB_SIZE - 1)


def warp_affine_cubic(src, mat, w_out, h_out, border=0.0):
    """cv2.warpAffine(src, mat, (w_out, h_out), flags=cv2.INTER_CUBIC, borderValue=border) for a float32 [H, W] image"""
    src = src.astype(np.float32)
    H, W = src.shape
    sx, sy, fx, fy = warp_coords(mat, w_out, h_out)
    tab = cubic_table()
    out = np.zeros((h_out, w_out), np.float32)
    for ky in range(4):
        yy = sy + ky
        for kx in range(4):
            xx = sx + kx
            inside = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            v = np.where(inside, src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], np.float32(border)).astype(np.float32)
            wgt = (tab[fy, ky] * tab[fx, kx]).astype(np.float32)         # the table holds float products
            out = (out + v * wgt).astype(np.float32)                     # accumulated row by row, left to right
    return out


def iou(pred, mask, thr=0.35):
    p = pred > thr
    m = mask.astype(bool)
    inter, union = np.logical_and(p, m).sum(), np.logical_or(p, m).sum()
    return float(inter) / (float(union) + 1e-6), int(inter), int(union)


def postprocess_one(logits, in_size, mat, ori_size, mask):
    """one sample of engine.py:100-123: logits [h, w] -> IoU against mask [ori_h, ori_w] (0/1)"""
    p = sigmoid(logits)
    if p.shape != tuple(in_size):
        p = upsample_bicubic(p, in_size[0], in_size[1])
    h, w = int(ori_size[0]), int(ori_size[1])
    p = warp_affine_cubic(p, mat, w, h, 0.0)
    return iou(p, mask)
