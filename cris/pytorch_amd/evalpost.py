"""Evaluation post-processing on the GPU - the per-batch body of the reference's `validate` / `inference`
(reference engine/engine.py:100-123, :171-188) behind the same quantities: logits in, per-sample IoU out.

    preds = torch.sigmoid(model(imgs, texts))                                    engine.py:100-101
    preds = F.interpolate(preds, size=imgs.shape[-2:], mode='bicubic', align_corners=True)          :102-106
    pred  = cv2.warpAffine(pred, mat, (w, h), flags=cv2.INTER_CUBIC, borderValue=0.)                :114-116
    pred  = pred > 0.35 ; iou = sum(pred & mask) / (sum(pred | mask) + 1e-6)                        :117-123

The reference copies every prediction to the host and warps it with cv2; here all four steps are kernels of
libcris_hip.so (csrc/evalpost.hip) and only the two integer counts per sample come back.  There is no CPU fallback."""
import ctypes as C

import numpy as np
import torch

from . import hip
from .hip import ptr


def _stream():
    return torch.cuda.current_stream().cuda_stream


def sigmoid_upsample(logits, H, W):
    """[B, 1, h, w] (or [B, h, w]) fp32 logits -> [B, H, W] probabilities: sigmoid + bicubic, align_corners=True"""
    if logits.device.type != "cuda":
        raise RuntimeError("evalpost runs on the GPU only (no CPU fallback)")
    x = logits.detach().float().contiguous()
    if x.dim() == 4:
        x = x[:, 0].contiguous()
    B, h, w = x.shape
    out = torch.empty(B, H, W, dtype=torch.float32, device=x.device)
    hip.call("cris_sigmoid_bicubic_up", ptr(x), B, h, w, H, W, ptr(out), _stream())
    return out


def warp_to_original(prob, mat, ori_size, border=0.0):
    """cv2.warpAffine(prob, mat, (w, h), flags=cv2.INTER_CUBIC, borderValue=border): prob [H, W] cuda fp32, mat the 2x3 matrix the
    reference passes (param['inverse'], utils/dataset.py:190-205), ori_size = (h, w) -> [h, w] cuda fp32"""
    H, W = prob.shape
    h, w = int(ori_size[0]), int(ori_size[1])
    m = np.ascontiguousarray(np.asarray(mat, dtype=np.float64).reshape(6))
    out = torch.empty(h, w, dtype=torch.float32, device=prob.device)
    hip.call("cris_warp_affine_cubic", ptr(prob.contiguous()), H, W, m.ctypes.data_as(C.c_void_p), w, h, float(border), ptr(out), _stream())
    return out


def iou_counts(pred, mask, thr=0.35):
    """(intersection, union) of (pred > thr) and (mask != 0) as device int32[2] (+=); pred, mask: same-shape cuda fp32"""
    counts = torch.zeros(2, dtype=torch.int32, device=pred.device)
    hip.call("cris_threshold_iou", ptr(pred.contiguous()), ptr(mask.contiguous()), pred.numel(), float(thr), ptr(counts), _stream())
    return counts


def validate_batch(logits, in_size, mats, ori_sizes, masks, thr=0.35):
    """One batch of engine.validate's loop (engine.py:100-123).  logits: model(imgs, texts) [B, 1, h, w]; in_size:
    imgs.shape[-2:]; mats / ori_sizes: param['inverse'] / param['ori_size'] per sample; masks: per-sample [ori_h, ori_w]
    tensors or arrays holding mask / 255.  Returns the list of per-sample IoUs (python floats, one host read for all)."""
    probs = sigmoid_upsample(logits, int(in_size[0]), int(in_size[1]))
    counts = []
    for b in range(probs.shape[0]):
        p = warp_to_original(probs[b], mats[b], ori_sizes[b])
        m = torch.as_tensor(np.asarray(masks[b], dtype=np.float32) if not torch.is_tensor(masks[b]) else masks[b],
                            dtype=torch.float32, device=p.device)
        counts.append(iou_counts(p, m, thr))
    c = torch.stack(counts).cpu().numpy().astype(np.float64)
    return [float(i / (u + 1e-6)) for i, u in c]
