"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's per-sample input preprocessing
(reference utils/dataset.py `RefDataset.__getitem__` :146-168, `getTransformMat` :190-205, `convert` :207-221):

    mat, mat_inv = getTransformMat(img.shape[:2], inverse=True)                   letterbox: scale = min(S/h, S/w), centred
    img  = cv2.warpAffine(img_rgb_u8, mat, (S, S), flags=cv2.INTER_CUBIC, borderValue=[0.48145466*255, 0.4578275*255, 0.40821073*255])
    mask = cv2.warpAffine(mask_u8,   mat, (S, S), flags=cv2.INTER_LINEAR, borderValue=0.) / 255.
    img  = (img.transpose(2, 0, 1).float() / 255 - mean) / std ;  mask = mask.float()

PINNING.  `convert` is torch / numpy arithmetic and is restated exactly (float32 ops in the same order; the mask goes
through float64 like `mask / 255.` does).  The two warps live in OpenCV (cv2 4.x, modules/imgproc/src/imgwarp.cpp), which is
neither in /root/reference (requirement.txt: `opencv-python`, unpinned) nor installed here: `warp_affine_u8` restates its
published 8-bit algorithm - destination coordinates in fixed point (AB_BITS 10) quantised to 1/32 pixel exactly as in
oracle/eval_post.py, then `remap` with the 16-bit weight tables of `initInterTab2D(fixpt=true)` (float separable weights
x 2^15, rounded, their sum pulled to 2^15 by correcting one entry - including that routine's quirk of searching the 2x2
window that STARTS at tap ksize/2, which for the bilinear table reaches into the following, not yet initialised entry), taps
outside the image read the border colour,
result `(sum + 2^14) >> 15` saturated to 8 bits - and is **parity unpinned** against cv2 itself (round 6: an independent sampler with
the same kernel and conventions, torch's F.grid_sample, agrees to 0.7 gray levels on a smooth image under the letter-box matrix -
tests/test_input_pipe.py - which pins the algorithm class, not cv2's bits).  `get_affine_transform`
solves the same 3-point system as cv2.getAffineTransform but with numpy's LU instead of OpenCV's SVD, so the matrix can
differ in the last bits.  Anchors asserted in tests/test_input_pipe.py: identity and integer translations copy pixels
exactly, weights are non-negative for INTER_LINEAR and every table row sums to 2^15, a constant image stays constant,
the letterbox matrix maps the image corners to the padded rectangle and `mat_inv` undoes `mat`.
Out of scope here (CPU-side in the reference's DataLoader workers): LMDB / pyarrow record
reading, JPEG / PNG decoding, the BPE tokenizer.
"""
import numpy as np

from .eval_post import AB_BITS, AB_SCALE, INTER_BITS, INTER_TAB_SIZE, invert_affine

INTER_LINEAR, INTER_CUBIC = 1, 2
COEF_BITS = 15
COEF_SCALE = 1 << COEF_BITS

MEAN = np.array([0.48145466, 0.4578275, 0.40821073], np.float32)       # utils/dataset.py:106-107
STD = np.array([0.26862954, 0.26130258, 0.27577711], np.float32)        # utils/dataset.py:108-109
BORDER_RGB = (0.48145466 * 255, 0.4578275 * 255, 0.40821073 * 255)       # utils/dataset.py:152


def get_affine_transform(src, dst):
    """cv2.getAffineTransform(src, dst): the 2x3 matrix with M @ [x, y, 1] = dst for three float32 point pairs (double result)"""
    src = np.asarray(src, np.float32).astype(np.float64)
    dst = np.asarray(dst, np.float32).astype(np.float64)
    A = np.concatenate([src, np.ones((3, 1))], 1)
    return np.linalg.solve(A, dst).T.copy()                  # rows: [a11 a12 b1], [a21 a22 b2]


def get_transform_mat(img_size, input_size, inverse=True):
    """RefDataset.getTransformMat (utils/dataset.py:190-205): aspect-preserving resize to the input size, centred"""
    ori_h, ori_w = img_size
    inp_h, inp_w = input_size
    scale = min(inp_h / ori_h, inp_w / ori_w)
    new_h, new_w = ori_h * scale, ori_w * scale
    bias_x, bias_y = (inp_w - new_w) / 2., (inp_h - new_h) / 2.
    src = np.array([[0, 0], [ori_w, 0], [0, ori_h]], np.float32)
    dst = np.array([[bias_x, bias_y], [new_w + bias_x, bias_y], [bias_x, new_h + bias_y]], np.float32)
    mat = get_affine_transform(src, dst)
    return (mat, get_affine_transform(dst, src)) if inverse else (mat, None)


def _coeffs_1d(method):
    """initInterTab1D: [32][ksize] float32 separable weights"""
    t = (np.arange(INTER_TAB_SIZE, dtype=np.float32) * np.float32(1.0 / INTER_TAB_SIZE)).astype(np.float32)
    if method == INTER_LINEAR:
        return np.stack([np.float32(1) - t, t], -1).astype(np.float32)
    A = np.float32(-0.75)
    c0 = ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A
    c1 = ((A + 2) * t - (A + 3)) * t * t + 1
    c2 = ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1
    c3 = np.float32(1.0) - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], -1).astype(np.float32)


def remap_table_u8(method):
    """initInterTab2D(method, fixpt=true): int16 [32*32][ksize*ksize] weights, entry (fy*32 + fx), tap (ky*ksize + kx)"""
    c = _coeffs_1d(method)
    ks = c.shape[1]
    n = INTER_TAB_SIZE * INTER_TAB_SIZE
    flat = np.zeros(n * ks * ks + 4 * ks + 4, np.int16)      # (the sum correction of the last entries peeks past them)
    for i in range(INTER_TAB_SIZE):
        for j in range(INTER_TAB_SIZE):
            base = (i * INTER_TAB_SIZE + j) * ks * ks
            v = (c[i][:, None] * c[j][None, :]).astype(np.float32) * np.float32(COEF_SCALE)
            it = np.clip(np.rint(v), -32768, 32767).astype(np.int16)          # saturate_cast<short>(float): cvRound, saturated
            flat[base:base + ks * ks] = it.reshape(-1)
            isum = int(it.astype(np.int64).sum())
            if isum != COEF_SCALE:
                diff = isum - COEF_SCALE
                k0 = ks // 2
                Mk = mk = (k0, k0)
                for k1 in range(k0, k0 + 2):
                    for k2 in range(k0, k0 + 2):
                        val = flat[base + k1 * ks + k2]
                        if val < flat[base + mk[0] * ks + mk[1]]:
                            mk = (k1, k2)
                        elif val > flat[base + Mk[0] * ks + Mk[1]]:
                            Mk = (k1, k2)
                tgt = Mk if diff < 0 else mk
                pos = base + tgt[0] * ks + tgt[1]
                flat[pos] = np.int16(np.int64(flat[pos]) - diff)
    return flat[:n * ks * ks].reshape(n, ks * ks).copy()


_TABLES = {}


def _table(method):
    if method not in _TABLES:
        _TABLES[method] = remap_table_u8(method)
    return _TABLES[method]


def warp_coords_u8(mat, w_out, h_out):
    """integer base pixel (X >> 5, Y >> 5) and table entry fy*32 + fx per destination pixel (cv::WarpAffineInvoker)"""
    M = invert_affine(mat)
    x = np.arange(w_out, dtype=np.float64)
    y = np.arange(h_out, dtype=np.float64)
    rnd = lambda v: np.rint(v).astype(np.int64)
    adelta, bdelta = rnd(M[0, 0] * x * AB_SCALE), rnd(M[1, 0] * x * AB_SCALE)
    round_delta = AB_SCALE // INTER_TAB_SIZE // 2
    X0 = rnd((M[0, 1] * y + M[0, 2]) * AB_SCALE) + round_delta
    Y0 = rnd((M[1, 1] * y + M[1, 2]) * AB_SCALE) + round_delta
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    return X >> INTER_BITS, Y >> INTER_BITS, (Y & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (X & (INTER_TAB_SIZE - 1))


def border_u8(border, C):
    """Scalar borderValue -> uint8 per channel (saturate_cast<uchar>(double): round half to even, clamp)"""
    b = np.zeros(C, np.float64)
    bb = np.atleast_1d(np.asarray(border, np.float64))
    b[:min(C, bb.size)] = bb[:C]
    return np.clip(np.rint(b), 0, 255).astype(np.uint8)


def warp_affine_u8(src, mat, w_out, h_out, method, border=0.0):
    """cv2.warpAffine(src_u8, mat, (w_out, h_out), flags=method, borderValue=border) for [H, W] or [H, W, C] uint8 images"""
    squeeze = src.ndim == 2
    s = src[:, :, None] if squeeze else src
    H, W, C = s.shape
    bx, by, ent = warp_coords_u8(mat, w_out, h_out)
    tab = _table(method).astype(np.int64)
    ks = 2 if method == INTER_LINEAR else 4
    off = 0 if method == INTER_LINEAR else -1
    cv = border_u8(border, C).astype(np.int64)
    acc = np.zeros((h_out, w_out, C), np.int64)
    for ky in range(ks):
        yy = by + off + ky
        for kx in range(ks):
            xx = bx + off + kx
            inside = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            v = np.where(inside[:, :, None], s[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64), cv[None, None, :])
            acc += v * tab[ent, ky * ks + kx][:, :, None]
    out = np.clip((acc + (1 << (COEF_BITS - 1))) >> COEF_BITS, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def convert_image(img_u8_hwc):
    """RefDataset.convert (utils/dataset.py:207-213): HWC uint8 -> CHW float32, /255, -mean, /std (float32 ops, in this order)"""
    x = img_u8_hwc.transpose(2, 0, 1).astype(np.float32)
    x = (x / np.float32(255.)).astype(np.float32)
    x = (x - MEAN[:, None, None]).astype(np.float32)
    return (x / STD[:, None, None]).astype(np.float32)


def convert_mask(mask_u8):
    """`mask / 255.` (float64) then `.float()` (utils/dataset.py:160, :216-219)"""
    return (mask_u8.astype(np.float64) / 255.).astype(np.float32)


def preprocess_train(img_rgb_u8, mask_u8, input_size):
    """one training sample (utils/dataset.py:146-163): -> img [3, S, S] f32, mask [S, S] f32, mat, mat_inv"""
    mat, mat_inv = get_transform_mat(img_rgb_u8.shape[:2], input_size, True)
    S_h, S_w = input_size
    img = warp_affine_u8(img_rgb_u8, mat, S_w, S_h, INTER_CUBIC, BORDER_RGB)
    out_mask = None
    if mask_u8 is not None:
        out_mask = convert_mask(warp_affine_u8(mask_u8, mat, S_w, S_h, INTER_LINEAR, 0.))
    return convert_image(img), out_mask, mat, mat_inv
