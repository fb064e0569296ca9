"""Input preprocessing (reference utils/dataset.py:146-168, :190-221): the numpy oracle's anchors,
the host-side tables of the library against the oracle (CPU), and the batch kernel against the oracle bit for bit (GPU)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import input_pipe as ip                     # noqa: E402
from oracle.eval_post import invert_affine              # noqa: E402


def _img(h, w, seed):
    rng = np.random.default_rng(seed)
    # smooth structure + noise so that cubic overshoot and saturation both occur
    yy, xx = np.mgrid[0:h, 0:w]
    base = 127 + 120 * np.sin(yy / 7.0)[:, :, None] * np.cos(xx[:, :, None] / 5.0 + np.arange(3)[None, None, :])
    return np.clip(base + rng.integers(-40, 40, (h, w, 3)), 0, 255).astype(np.uint8)


def _mask(h, w, seed):
    rng = np.random.default_rng(seed)
    m = np.zeros((h, w), np.uint8)
    y0, x0 = rng.integers(0, h // 2), rng.integers(0, w // 2)
    m[y0:y0 + h // 3 + 1, x0:x0 + w // 3 + 1] = 255
    return m


def test_tables_rows_sum_to_one_in_fixed_point():
    lin, cub = ip.remap_table_u8(ip.INTER_LINEAR), ip.remap_table_u8(ip.INTER_CUBIC)
    assert lin.shape == (1024, 4) and cub.shape == (1024, 16)
    assert (lin.astype(np.int64).sum(1) == 32768).all() and (cub.astype(np.int64).sum(1) == 32768).all()
    assert (lin >= 0).all()
    assert lin[0].tolist() == [32767, 0, 0, 1]            # 2^15 does not fit a short: the correction lands on the last tap
    assert cub[0].reshape(4, 4)[1, 1] == 32767 and cub[0].astype(np.int64).sum() == 32768
    # half-pixel entry of the bilinear table: four equal quarters
    assert lin[16 * 32 + 16].tolist() == [8192, 8192, 8192, 8192]


def test_identity_and_integer_translation_copy_pixels():
    img, msk = _img(37, 53, 0), _mask(37, 53, 0)
    eye = np.array([[1, 0, 0], [0, 1, 0]], np.float64)
    for method in (ip.INTER_LINEAR, ip.INTER_CUBIC):
        assert np.array_equal(ip.warp_affine_u8(img, eye, 53, 37, method, 0.), img)
        assert np.array_equal(ip.warp_affine_u8(msk, eye, 53, 37, method, 0.), msk)
        sh = np.array([[1, 0, 5], [0, 1, -3]], np.float64)   # dst(x, y) = src(x - 5, y + 3)
        out = ip.warp_affine_u8(img, sh, 53, 37, method, (9, 8, 7))
        assert np.array_equal(out[0:34, 5:53], img[3:37, 0:48])
        assert (out[:, :5] == np.array([9, 8, 7], np.uint8)).all() and (out[34:] == np.array([9, 8, 7], np.uint8)).all()


def test_constant_image_stays_constant_and_border_colour_is_rounded():
    img = np.full((40, 60, 3), 200, np.uint8)
    mat, _ = ip.get_transform_mat((40, 60), (96, 96), True)
    out = ip.warp_affine_u8(img, mat, 96, 96, ip.INTER_CUBIC, ip.BORDER_RGB)
    assert ip.border_u8(ip.BORDER_RGB, 3).tolist() == [123, 117, 104]
    inside = out[22:74, 6:90]            # (more than two source pixels away from the image edge)
    assert (inside == 200).all()
    assert (out[0:10] == np.array([123, 117, 104], np.uint8)).all()


def test_u8_warps_against_an_independent_sampler():
    """beside the restatement of OpenCV's 8-bit warps, an independent implementation of the same sampling rules (round 6; does
    not lift "parity unpinned" - cv2 stays absent - but pins the algorithm class): torch's F.grid_sample with the Keys kernel
    A = -0.75 (`bicubic`) / `bilinear`, integer pixel coordinates (align_corners=True), on exact coordinates in float.  OpenCV
    quantises coordinates to 1/32 pixel and weights to 15 bits and rounds to 8 bits: on a smooth image under the dataset's
    letter-box matrix (`utils/dataset.py:190-205`) the two agree to a gray level or two in the interior."""
    import torch.nn.functional as F
    h, w, S = 97, 141, 160
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127 + 100 * np.sin(yy / 9.0 + c) * np.cos(xx / 11.0) for c in range(3)], -1)
    img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    mask = (((yy - 40) ** 2 / 900.0 + (xx - 70) ** 2 / 2500.0) < 1.0).astype(np.uint8) * 255
    mat, _ = ip.get_transform_mat((h, w), (S, S), True)
    A = np.vstack([mat, [0, 0, 1]])
    Ai = np.linalg.inv(A)
    oy, ox = np.mgrid[0:S, 0:S].astype(np.float64)
    sx = Ai[0, 0] * ox + Ai[0, 1] * oy + Ai[0, 2]
    sy = Ai[1, 0] * ox + Ai[1, 1] * oy + Ai[1, 2]
    grid = torch.from_numpy(np.stack([2 * sx / (w - 1) - 1, 2 * sy / (h - 1) - 1], -1)[None]).float()
    inside = (sx >= 2) & (sx <= w - 3) & (sy >= 2) & (sy <= h - 3)
    got = ip.warp_affine_u8(img, mat, S, S, ip.INTER_CUBIC, ip.BORDER_RGB).astype(np.float64)
    ref = F.grid_sample(torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None], grid, mode="bicubic", padding_mode="zeros",
                        align_corners=True)[0].permute(1, 2, 0).numpy().clip(0, 255)
    d = np.abs(got - ref)[inside]
    assert d.max() <= 1.5 and d.mean() <= 0.4, (d.max(), d.mean())          # measured: max 0.69, mean 0.26 gray levels (rounding to 8 bits)
    gotm = ip.warp_affine_u8(mask, mat, S, S, ip.INTER_LINEAR, 0.).astype(np.float64)
    refm = F.grid_sample(torch.from_numpy(mask.astype(np.float32))[None, None], grid, mode="bilinear", padding_mode="zeros",
                         align_corners=True)[0, 0].numpy()
    dm = np.abs(gotm - refm)[inside]
    # (a 0 / 255 step edge: 1/32 pixel of coordinate quantisation is up to 255 / 32 = 8 gray levels ON the edge, nothing elsewhere)
    assert dm.max() <= 255.0 / 32.0 + 1.0 and dm.mean() <= 0.1, (dm.max(), dm.mean())        # measured: max 5.3 (on the edge), mean 0.044


def test_letterbox_matrix():
    for (h, w), S in (((480, 640), 416), ((640, 427), 416), ((50, 37), 96), ((416, 416), 416)):
        mat, inv = ip.get_transform_mat((h, w), (S, S), True)
        scale = min(S / h, S / w)
        nh, nw = h * scale, w * scale
        corners = np.array([[0, 0, 1], [w, 0, 1], [0, h, 1]], np.float64)
        want = np.array([[(S - nw) / 2, (S - nh) / 2], [(S - nw) / 2 + nw, (S - nh) / 2], [(S - nw) / 2, (S - nh) / 2 + nh]])
        assert np.allclose(corners @ mat.T, want, atol=1e-4)
        full = np.vstack([mat, [0, 0, 1]]) @ np.vstack([inv, [0, 0, 1]])
        assert np.allclose(full, np.eye(3), atol=1e-5)


def test_convert_equals_the_reference_tensor_ops():
    """RefDataset.convert (utils/dataset.py:207-221) with torch itself"""
    img, msk = _img(64, 64, 3), _mask(64, 64, 3)
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).reshape(3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).reshape(3, 1, 1)
    t = torch.from_numpy(img.transpose((2, 0, 1))).float()
    t.div_(255.).sub_(mean).div_(std)
    assert np.array_equal(ip.convert_image(img), t.numpy())
    m = torch.from_numpy(msk / 255.).float()
    assert np.array_equal(ip.convert_mask(msk), m.numpy())


def test_library_host_tables_match_the_oracle():
    from cris.pytorch_amd import hip
    lin, cub = np.zeros(1024 * 4, np.int16), np.zeros(1024 * 16, np.int16)
    hip.call("cris_remap_tables_u8", lin.ctypes.data_as(C.c_void_p), cub.ctypes.data_as(C.c_void_p))
    assert np.array_equal(lin.reshape(1024, 4), ip.remap_table_u8(ip.INTER_LINEAR))
    assert np.array_equal(cub.reshape(1024, 16), ip.remap_table_u8(ip.INTER_CUBIC))
    mat, _ = ip.get_transform_mat((480, 640), (416, 416), True)
    m, inv = np.ascontiguousarray(mat.reshape(6)), np.zeros(6, np.float64)
    hip.call("cris_invert_affine", m.ctypes.data_as(C.c_void_p), inv.ctypes.data_as(C.c_void_p))
    assert np.array_equal(inv.reshape(2, 3), invert_affine(mat))
    from cris.pytorch_amd import inputpipe
    for size in ((480, 640), (333, 500), (50, 37)):
        a, b = inputpipe.get_transform_mat(size, (416, 416), True), ip.get_transform_mat(size, (416, 416), True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.gpu
def test_preprocess_batch_bit_exact_vs_oracle():
    """ragged batch (landscape, portrait, tiny image that is enlarged, exact-size image, one sample without a mask)"""
    from cris.pytorch_amd.inputpipe import Preprocessor
    sizes = [(480, 640), (640, 427), (50, 37), (416, 416), (333, 500)]
    for S in (416, 480):
        pre = Preprocessor((S, S))
        imgs = [_img(h, w, i) for i, (h, w) in enumerate(sizes)]
        masks = [_mask(h, w, i) if i != 2 else None for i, (h, w) in enumerate(sizes)]
        img, mask, mats, invs = pre(imgs, masks)
        torch.cuda.synchronize()
        for i in range(len(sizes)):
            want_img, want_mask, mat, inv = ip.preprocess_train(imgs[i], masks[i], (S, S))
            assert np.array_equal(mats[i], mat) and np.array_equal(invs[i], inv)
            got = img[i].cpu().numpy()
            assert np.array_equal(got, want_img), (S, i, np.abs(got - want_img).max(), int((got != want_img).sum()))
            if masks[i] is not None:
                assert np.array_equal(mask[i].cpu().numpy(), want_mask), (S, i)
            else:
                assert float(mask[i].abs().max()) == 0.0
