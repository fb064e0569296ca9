"""Input preprocessing on the GPU - the array part of the reference's `RefDataset.__getitem__`
(reference utils/dataset.py:146-168, `getTransformMat` :190-205, `convert` :207-221) for a whole batch in one launch:

    mat, mat_inv = self.getTransformMat(img_size, True)
    img  = cv2.warpAffine(img, mat, self.input_size, flags=cv2.INTER_CUBIC, borderValue=[0.48145466*255, 0.4578275*255, 0.40821073*255])
    mask = cv2.warpAffine(mask, mat, self.input_size, flags=cv2.INTER_LINEAR, borderValue=0.) / 255.
    img, mask = self.convert(img, mask)          # CHW float, /255, -mean, /std

Inputs are the DECODED uint8 arrays (RGB [h, w, 3], mask [h, w]); record reading, JPEG / PNG decoding and the tokenizer stay
on the CPU side of the loader.  The kernels are in csrc/inputpipe.hip; there is no CPU
fallback.  The 8-bit warp arithmetic is OpenCV's, restated (oracle/input_pipe.py explains what that is anchored on)."""
import ctypes as C

import numpy as np
import torch

from . import hip
from .hip import ptr

MEAN = (0.48145466, 0.4578275, 0.40821073)      # utils/dataset.py:106-107
STD = (0.26862954, 0.26130258, 0.27577711)       # utils/dataset.py:108-109


def get_affine_transform(src, dst):
    """cv2.getAffineTransform: 2x3 double matrix mapping three float32 points src -> dst"""
    src = np.asarray(src, np.float32).astype(np.float64)
    dst = np.asarray(dst, np.float32).astype(np.float64)
    return np.linalg.solve(np.concatenate([src, np.ones((3, 1))], 1), dst).T.copy()


def get_transform_mat(img_size, input_size, inverse=False):
    """RefDataset.getTransformMat (utils/dataset.py:190-205): same arguments, same return (mat, mat_inv or None)"""
    ori_h, ori_w = img_size
    inp_h, inp_w = input_size
    scale = min(inp_h / ori_h, inp_w / ori_w)
    new_h, new_w = ori_h * scale, ori_w * scale
    bias_x, bias_y = (inp_w - new_w) / 2., (inp_h - new_h) / 2.
    src = np.array([[0, 0], [ori_w, 0], [0, ori_h]], np.float32)
    dst = np.array([[bias_x, bias_y], [new_w + bias_x, bias_y], [bias_x, new_h + bias_y]], np.float32)
    mat = get_affine_transform(src, dst)
    return (mat, get_affine_transform(dst, src)) if inverse else (mat, None)


class Preprocessor:
    """Device-resident tables (remap weights, normalisation look-ups) + `__call__` for one batch."""

    def __init__(self, input_size, device="cuda:0"):
        self.input_size = (int(input_size[0]), int(input_size[1]))
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("the input pipeline runs on the GPU only (no CPU fallback)")
        lin, cub = np.zeros(1024 * 4, np.int16), np.zeros(1024 * 16, np.int16)
        hip.call("cris_remap_tables_u8", lin.ctypes.data_as(C.c_void_p), cub.ctypes.data_as(C.c_void_p))
        self.tab_linear, self.tab_cubic = torch.from_numpy(lin).to(self.dev), torch.from_numpy(cub).to(self.dev)
        v = np.arange(256, dtype=np.float32)
        mean, std = np.array(MEAN, np.float32), np.array(STD, np.float32)
        # img.float().div_(255.).sub_(mean).div_(std): three float32 operations per value, in this order
        lut = (((v[None, :] / np.float32(255.)).astype(np.float32) - mean[:, None]).astype(np.float32) / std[:, None]).astype(np.float32)
        self.lut_img = torch.from_numpy(np.ascontiguousarray(lut)).to(self.dev)
        # mask / 255. is a float64 division in numpy, then .float()
        self.lut_mask = torch.from_numpy((np.arange(256, dtype=np.float64) / 255.).astype(np.float32)).to(self.dev)
        # cv2 turns the Scalar border colour into pixel type: saturate_cast<uchar>(double)
        self.border = np.clip(np.rint(np.array(MEAN, np.float64) * 255), 0, 255).astype(np.uint8)

    def __call__(self, images, masks=None):
        """images: list of uint8 RGB arrays / tensors [h, w, 3]; masks: optional list of uint8 [h, w] (None entries allowed).
        Returns (img [B, 3, S, S] float32, mask [B, S, S] float32 or None, mats, mat_invs) - mats as getTransformMat gives them."""
        B = len(images)
        S_h, S_w = self.input_size
        keep, descs, mats, invs = [], (hip.SampleDesc * B)(), [], []
        for i, im in enumerate(images):
            t = torch.as_tensor(im)
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
                raise ValueError("image %d must be uint8 [h, w, 3], got %s %s" % (i, t.dtype, tuple(t.shape)))
            t = t.to(self.dev, non_blocking=True).contiguous()
            keep.append(t)
            h, w = int(t.shape[0]), int(t.shape[1])
            mat, mat_inv = get_transform_mat((h, w), self.input_size, True)
            mats.append(mat)
            invs.append(mat_inv)
            d = descs[i]
            d.img, d.H, d.W = t.data_ptr(), h, w
            m = np.ascontiguousarray(mat.reshape(6))
            hip.call("cris_invert_affine", m.ctypes.data_as(C.c_void_p), C.addressof(d) + hip.SampleDesc.inv.offset)
            d.mask = None
            if masks is not None and masks[i] is not None:
                mk = torch.as_tensor(masks[i])
                if mk.dtype != torch.uint8 or tuple(mk.shape) != (h, w):
                    raise ValueError("mask %d must be uint8 [%d, %d], got %s %s" % (i, h, w, mk.dtype, tuple(mk.shape)))
                mk = mk.to(self.dev, non_blocking=True).contiguous()
                keep.append(mk)
                d.mask = mk.data_ptr()
        table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.dev)
        img = torch.empty(B, 3, S_h, S_w, dtype=torch.float32, device=self.dev)
        mask = torch.zeros(B, S_h, S_w, dtype=torch.float32, device=self.dev) if masks is not None else None
        hip.call("cris_preprocess_batch", ptr(table), B, S_h, S_w, ptr(self.tab_linear), ptr(self.tab_cubic), ptr(self.lut_img),
                 ptr(self.lut_mask), self.border.ctypes.data_as(C.c_void_p), ptr(img), ptr(mask), torch.cuda.current_stream().cuda_stream)
        self._keep = (keep, table)          # until the next call: the launch reads them asynchronously
        return img, mask, mats, invs
