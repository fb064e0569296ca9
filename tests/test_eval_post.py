"""Evaluation post-processing (reference engine/engine.py:100-123): the oracle restatement against torch (bicubic) and against
properties every correct affine warp has (cv2 itself is absent: the warp is "parity unpinned", see oracle/eval_post.py), on
the CPU; the HIP kernels against the oracle on the GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import eval_post as EP


def _rand(h, w, seed=0):
    return np.random.default_rng(seed).standard_normal((h, w)).astype(np.float32)


def test_oracle_bicubic_matches_torch():
    for (h, w, H, W) in ((26, 26, 104, 104), (13, 17, 52, 40), (104, 104, 416, 416), (120, 120, 480, 480)):
        x = _rand(h, w, h + w)
        ref = F.interpolate(torch.from_numpy(x)[None, None], size=(H, W), mode="bicubic", align_corners=True)[0, 0].numpy()
        got = EP.upsample_bicubic(x, H, W)
        assert np.abs(got - ref).max() < 5e-6, (h, w, H, W, np.abs(got - ref).max())      # summation order of 16 float products


def test_oracle_warp_identity_and_translation():
    src = _rand(37, 53, 1)
    ident = np.array([[1, 0, 0], [0, 1, 0]], np.float64)
    assert np.array_equal(EP.warp_affine_cubic(src, ident, 53, 37), src)         # fraction 0: weights (0, 1, 0, 0) exactly
    shift = np.array([[1, 0, 5], [0, 1, -3]], np.float64)                          # dst(x, y) = src(x - 5, y + 3)
    out = EP.warp_affine_cubic(src, shift, 53, 37)
    assert np.array_equal(out[0:34, 5:53], src[3:37, 0:48])
    assert np.all(out[:, :5] == 0) and np.all(out[34:, :] == 0)                   # borderValue 0 outside the source


def test_oracle_warp_cubic_table_and_half_pixel():
    tab = EP.cubic_table()
    assert tab.shape == (32, 4) and np.allclose(tab.sum(1), 1.0, atol=1e-6)
    assert np.array_equal(tab[0], np.array([0, 1, 0, 0], np.float32))
    assert np.allclose(tab[16], [-0.09375, 0.59375, 0.59375, -0.09375])           # Keys cubic, A = -0.75, at t = 1/2
    src = np.tile(np.arange(20, dtype=np.float32), (8, 1))                         # linear ramp: cubic interpolation is exact
    half = np.array([[1, 0, -0.5], [0, 1, 0]], np.float64)                         # dst(x) = src(x + 0.5)
    out = EP.warp_affine_cubic(src, half, 20, 8)
    assert np.allclose(out[:, 2:17], src[:, 2:17] + 0.5, atol=1e-5)


def test_oracle_warp_against_an_independent_bicubic_sampler():
    """A second, independent implementation of the same sampling rule beside the restatement (round 6; cv2 itself stays absent,
    so this does not lift "parity unpinned" - it pins the algorithm CLASS): torch's F.grid_sample(mode="bicubic",
    padding_mode="zeros", align_corners=True) uses the same Keys kernel (A = -0.75), integer pixel coordinates and zero border
    that cv2.warpAffine(INTER_CUBIC, borderValue=0) uses, in float arithmetic on exact coordinates; OpenCV quantises the source
    coordinate to 1/32 pixel and takes its weights from a table.  On a smooth image, under the matrices the dataset produces
    (letter-box inverse: scale + translation, `utils/dataset.py:190-205`) and under a rotation + anisotropic scale, the two must
    agree to the quantisation error: |d| <= max gradient x 1/64 pixel x (|x| + |y| parts) + rounding."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(5)
    H, W = 61, 83
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    src = (0.5 + 0.25 * np.sin(xx / 7.0) * np.cos(yy / 5.0) + 0.2 * np.sin((xx + yy) / 11.0)).astype(np.float32)      # smooth, in [0, 1]
    gmax = float(max(np.abs(np.diff(src, axis=0)).max(), np.abs(np.diff(src, axis=1)).max()))
    th = 0.3
    mats = [np.array([[1 / 0.832, 0, -10.0 / 0.832], [0, 1 / 0.832, -33.5 / 0.832]], np.float64),          # a letter-box inverse
            np.array([[1.7 * np.cos(th), -1.1 * np.sin(th), 6.3], [1.7 * np.sin(th), 1.1 * np.cos(th), -4.2]], np.float64),
            np.array([[0.61, 0.0, 3.37], [0.0, 0.61, 1.91]], np.float64)]
    for mat, (wo, ho) in zip(mats, ((97, 71), (120, 90), (50, 40))):
        got = EP.warp_affine_cubic(src, mat, wo, ho, 0.0)
        A = np.vstack([mat, [0, 0, 1]])
        Ai = np.linalg.inv(A)                                     # cv2.warpAffine without WARP_INVERSE_MAP: dst -> src through M^-1
        oy, ox = np.mgrid[0:ho, 0:wo].astype(np.float64)
        sx = Ai[0, 0] * ox + Ai[0, 1] * oy + Ai[0, 2]
        sy = Ai[1, 0] * ox + Ai[1, 1] * oy + Ai[1, 2]
        grid = torch.from_numpy(np.stack([2 * sx / (W - 1) - 1, 2 * sy / (H - 1) - 1], -1)[None]).float()
        ref = F.grid_sample(torch.from_numpy(src)[None, None], grid, mode="bicubic", padding_mode="zeros", align_corners=True)[0, 0].numpy()
        inside = (sx >= 2) & (sx <= W - 3) & (sy >= 2) & (sy <= H - 3)         # all 16 taps inside: the border rule aside
        d = np.abs(got - ref)
        bound = 2.0 * gmax / 64.0 * 1.3 + 1e-4                    # 1/64 pixel in x and in y, cubic overshoot 1.3
        assert d[inside].max() <= bound, (d[inside].max(), bound)
        assert d[inside].mean() <= 0.35 * bound
        # at the border both take zero for the taps outside the image: same rule, same order of magnitude of agreement
        assert d.max() <= 4.0 * bound + 0.02, d.max()
        # outside the source (beyond two pixels) both give exactly the border value
        far = (sx < -2) | (sx > W + 1) | (sy < -2) | (sy > H + 1)
        assert np.all(got[far] == 0.0) and np.all(ref[far] == 0.0)
    _ = rng


def test_oracle_dataset_matrix_pair_round_trip():
    """utils/dataset.py:190-205: an original 300 x 500 image letter-boxed into 416 x 416 and back: a mask that is constant
    inside the valid area comes back constant in the interior of the original frame"""
    ori_h, ori_w, S = 300, 500, 416
    scale = min(S / ori_h, S / ori_w)
    new_h, new_w = ori_h * scale, ori_w * scale
    bx, by = (S - new_w) / 2.0, (S - new_h) / 2.0
    mat_inv = np.array([[1 / scale, 0, -bx / scale], [0, 1 / scale, -by / scale]], np.float64)       # input -> original
    prob = np.zeros((S, S), np.float32)
    prob[int(by):int(by + new_h), :] = 1.0                                          # the letter-boxed image area
    out = EP.warp_affine_cubic(prob, mat_inv, ori_w, ori_h)
    assert out.shape == (ori_h, ori_w)
    assert np.allclose(out[4:-4, 4:-4], 1.0, atol=1e-5)
    iou, inter, union = EP.iou(out, np.ones((ori_h, ori_w), np.float32))
    assert iou > 0.98


@pytest.mark.gpu
def test_hip_eval_post_matches_oracle():
    from cris.pytorch_amd import evalpost
    dev = torch.device("cuda:0")
    B, h, S = 3, 104, 416
    logits = torch.from_numpy(np.stack([_rand(h, h, 10 + b) * 3 for b in range(B)]))[:, None]
    probs = evalpost.sigmoid_upsample(logits.to(dev), S, S).cpu().numpy()
    for b in range(B):
        ref = EP.upsample_bicubic(EP.sigmoid(logits[b, 0].numpy()), S, S)
        err = float(np.abs(probs[b] - ref).max())
        print("sigmoid + bicubic max abs err vs oracle: %.3e" % err)
        assert err < 2e-5, err                  # expf / fused multiply-adds in the weight polynomials vs numpy float32
    sizes = [(300, 500), (480, 640), (333, 251)]
    mats, masks = [], []
    for (oh, ow) in sizes:
        scale = min(S / oh, S / ow)
        bx, by = (S - ow * scale) / 2.0, (S - oh * scale) / 2.0
        mats.append(np.array([[1 / scale, 0, -bx / scale], [0, 1 / scale, -by / scale]], np.float64))
        masks.append((np.random.default_rng(oh).random((oh, ow)) > 0.6).astype(np.float32))
    # the warp alone, on identical inputs: the same arithmetic in the same order -> equal values
    for b, (oh, ow) in enumerate(sizes):
        got = evalpost.warp_to_original(torch.from_numpy(probs[b]).to(dev), mats[b], (oh, ow)).cpu().numpy()
        ref = EP.warp_affine_cubic(probs[b], mats[b], ow, oh)
        assert np.array_equal(got, ref), np.abs(got - ref).max()
    # a rotation + shear (general matrix), and the whole batch function against the oracle's IoU
    rot = np.array([[0.9, -0.3, 20.5], [0.25, 1.1, -7.25]], np.float64)
    got = evalpost.warp_to_original(torch.from_numpy(probs[0]).to(dev), rot, (200, 300)).cpu().numpy()
    assert np.array_equal(got, EP.warp_affine_cubic(probs[0], rot, 300, 200))
    ious = evalpost.validate_batch(logits.to(dev), (S, S), mats, sizes, masks)
    for b, (oh, ow) in enumerate(sizes):
        ref_iou, inter, union = EP.postprocess_one(logits[b, 0].numpy(), (S, S), mats[b], (oh, ow), masks[b])
        assert abs(ious[b] - ref_iou) < 2e-3, (b, ious[b], ref_iou)          # (a probability within 2e-5 of 0.35 may flip)
