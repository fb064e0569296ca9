"""TEST INFRASTRUCTURE ONLY - fp32 CPU restatement of the reference CRIS forward + loss.

This is the oracle the HIP path is checked against (tests/, __graft_entry__.smoke(), and the
`cpu_baseline` leg of bench.py).  Nothing in the product path (cris/pytorch_amd/) may import it.
It is device-generic functional torch: the pinned form is fp32 on the CPU; the same code on the GPU
(fp32 - checked equal to the CPU run in tests/test_oracle_device.py - or under torch.autocast, the
reference's own precision policy engine/engine.py:48) is what tools/parity_study.py and the
teacher-forced trajectory test use as "stock PyTorch on this hardware".

It restates, function by function, what the reference computes on the training hot path
(reference = DerrickWang005/CRIS.pytorch; citations are file:line in that repo), as plain
functional torch fp32 code over a state_dict with the reference's key names.  The arithmetic of
the reference itself lives in PyTorch (third-party; the reference pins no version - SURVEY.md
section 8c); this file uses the same torch primitives for convolution / interpolation and spells
out batch-norm, layer-norm, attention (incl. dropout on the probabilities) and the loss by hand
so that dropout masks can come from oracle/dropout_hash.py instead of torch's Philox stream.

Pinning: tests/golden/make_golden.py imports the real reference in the build container, loads the
same synthetic state_dict into it and stores its outputs / loss / gradients as fixtures;
tests/test_oracle_golden.py checks this file against those fixtures ("parity pinned by executing
the reference itself", dropout 0 - with dropout > 0 torch's RNG cannot be matched by anything).
Gradients of the oracle come from torch autograd over this forward.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import dropout_hash

BN_EPS = 1e-5
BN_MOM = 0.1
LN_EPS = 1e-5


# ----------------------------------------------------------------------------------------------
# dropout (HIP-path convention: keep decided per element index of a row-major [rows, cols] view)
# ----------------------------------------------------------------------------------------------
class DropCtx:
    def __init__(self, p: float, seed: Optional[int]):
        self.p = float(p)
        self.seed = seed

    @property
    def active(self):
        return self.seed is not None and self.p > 0.0

    def apply(self, x: torch.Tensor, stream: int) -> torch.Tensor:
        """x is already laid out in the HIP path's element order (row-major, contiguous)."""
        if not self.active:
            return x
        if x.device.type == "cpu":
            m = torch.from_numpy(dropout_hash.keep_mask(self.seed, stream, x.numel(), self.p)).view(x.shape).to(x.dtype)
        else:
            m = dropout_hash.keep_mask_torch(self.seed, stream, x.numel(), self.p, x.device).view(x.shape).to(x.dtype)
        return x * m * (1.0 / (1.0 - self.p))


def drop_stream(layer: int, site: int) -> int:
    """stream id of a dropout site; site: 0 self-attn probs, 1 dropout1, 2 cross-attn probs,
    3 dropout2, 4 ffn dropout, 5 dropout3 (reference model/layers.py:202-219)."""
    return layer * 8 + site


# ----------------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------------
# NATIVE_NORMS = True: BatchNorm / LayerNorm through torch's own F.batch_norm / F.layer_norm instead of the spelled-out forms
# below - the operators the reference's nn.BatchNorm2d / nn.LayerNorm modules call, so that under torch.autocast they follow
# autocast's own per-operator policy (tools/parity_study.py: "what does stock PyTorch autocast do on this network").
NATIVE_NORMS = False


def batch_norm(x, sd, prefix, training, bn_updates=None):
    """torch BatchNorm (eps 1e-5, momentum 0.1, biased var for normalisation, unbiased for the
    running estimate).  x: [B,C,H,W] or [B,C]."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if NATIVE_NORMS:
        rm = rv = None
        if bn_updates is not None or not training:
            rm, rv = sd[prefix + ".running_mean"].detach().clone(), sd[prefix + ".running_var"].detach().clone()
        y = F.batch_norm(x, rm, rv, w, b, training, BN_MOM, BN_EPS)
        if training and bn_updates is not None:
            bn_updates[prefix] = (rm, rv)
        return y
    dims = [0] + list(range(2, x.dim()))
    shape = [1, -1] + [1] * (x.dim() - 2)
    if training:
        mean = x.mean(dims)
        var = ((x - mean.view(shape)) ** 2).mean(dims)
        n = x.numel() // x.shape[1]
        if bn_updates is not None:
            rm, rv = sd[prefix + ".running_mean"].detach(), sd[prefix + ".running_var"].detach()
            bn_updates[prefix] = ((1 - BN_MOM) * rm + BN_MOM * mean.detach(),
                                  (1 - BN_MOM) * rv + BN_MOM * var.detach() * (n / max(n - 1, 1)))
    else:
        mean, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean.view(shape)) * (inv * w).view(shape) + b.view(shape)


def layer_norm(x, sd, prefix):
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if NATIVE_NORMS:
        return F.layer_norm(x, (x.shape[-1],), w, b, LN_EPS)
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + LN_EPS) * w + b


def mha_core(q, k, v, nheads, add_mask=None, key_pad=None, drop: Optional[DropCtx] = None, stream=0):
    """q [B,Lq,E], k,v [B,Lk,E] already projected.  torch MHA math path: q scaled by d**-0.5
    before QK^T, additive mask, softmax, dropout on probabilities, PV."""
    B, Lq, E = q.shape
    Lk = k.shape[1]
    d = E // nheads
    qh = q.view(B, Lq, nheads, d).transpose(1, 2) * (d ** -0.5)
    kh = k.view(B, Lk, nheads, d).transpose(1, 2)
    vh = v.view(B, Lk, nheads, d).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)                       # [B,H,Lq,Lk]
    if add_mask is not None:
        s = s + add_mask
    if key_pad is not None:
        s = s.masked_fill(key_pad[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    if drop is not None and drop.active:
        p = drop.apply(p.contiguous(), stream)          # idx = ((b*H+h)*Lq+q)*Lk+k
    o = p @ vh
    return o.transpose(1, 2).reshape(B, Lq, E)


# ----------------------------------------------------------------------------------------------
# CLIP visual encoder   (reference model/clip.py:10-223)
# ----------------------------------------------------------------------------------------------
def bottleneck(x, sd, p, stride, training, bnu):
    # reference model/clip.py:44-57
    out = F.relu(batch_norm(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1", training, bnu))
    out = F.relu(batch_norm(F.conv2d(out, sd[p + ".conv2.weight"], padding=1), sd, p + ".bn2", training, bnu))
    if stride > 1:
        out = F.avg_pool2d(out, stride)
    out = batch_norm(F.conv2d(out, sd[p + ".conv3.weight"]), sd, p + ".bn3", training, bnu)
    if (p + ".downsample.0.weight") in sd:
        idt = F.avg_pool2d(x, stride) if stride > 1 else x
        idt = batch_norm(F.conv2d(idt, sd[p + ".downsample.0.weight"]), sd, p + ".downsample.1", training, bnu)
    else:
        idt = x
    return F.relu(out + idt)


def attnpool(x, sd, p, nheads, grid, training, bnu):
    # reference model/clip.py:110-144 (all HW tokens are queries; residual `connect` branch)
    B, C, H, W = x.shape
    res = batch_norm(F.conv2d(x, sd[p + ".connect.0.weight"]), sd, p + ".connect.1", training, bnu)
    pos = sd[p + ".positional_embedding"]                               # [grid*grid+1, C]
    pw = pos[1:].reshape(1, grid, grid, C).permute(0, 3, 1, 2)
    pw = F.interpolate(pw, size=(H, W), align_corners=False, mode="bicubic")  # clip.py:101-104
    t = x.reshape(B, C, H * W) + pw.flatten(2)                          # NC(HW)
    t = t.permute(0, 2, 1)                                              # [B, HW, C]
    q = F.linear(t, sd[p + ".q_proj.weight"], sd[p + ".q_proj.bias"])
    k = F.linear(t, sd[p + ".k_proj.weight"], sd[p + ".k_proj.bias"])
    v = F.linear(t, sd[p + ".v_proj.weight"], sd[p + ".v_proj.bias"])
    o = mha_core(q, k, v, nheads)
    o = F.linear(o, sd[p + ".c_proj.weight"], sd[p + ".c_proj.bias"])  # [B, HW, Cout]
    o = o.permute(0, 2, 1).reshape(B, -1, H, W)
    return F.relu(o + res)


def encode_image(img, sd, clip, training, bnu, taps=None):
    # reference model/clip.py:207-223
    v = "backbone.visual"
    x = img
    for i, stride in ((1, 2), (2, 1), (3, 1)):
        x = F.conv2d(x, sd["%s.conv%d.weight" % (v, i)], stride=stride, padding=1)
        x = F.relu(batch_norm(x, sd, "%s.bn%d" % (v, i), training, bnu))
    x = F.avg_pool2d(x, 2)
    if taps is not None:
        taps["stem"] = x
    feats = []
    for li, nblk in enumerate(clip.vision_layers):
        for bi in range(nblk):
            stride = 2 if (li > 0 and bi == 0) else 1
            x = bottleneck(x, sd, "%s.layer%d.%d" % (v, li + 1, bi), stride, training, bnu)
        feats.append(x)
        if taps is not None:
            taps["layer%d" % (li + 1)] = x
    x4 = attnpool(feats[3], sd, v + ".attnpool", clip.vis_heads, clip.pos_grid, training, bnu)
    if taps is not None:
        taps["attnpool"] = x4
    return feats[1], feats[2], x4


# ----------------------------------------------------------------------------------------------
# CLIP text encoder   (reference model/clip.py:226-283, 424-456)
# ----------------------------------------------------------------------------------------------
def encode_text(word, sd, clip, taps=None):
    B, L = word.shape
    x = sd["backbone.token_embedding.weight"][word] + sd["backbone.positional_embedding"][:L]
    causal = torch.full((L, L), float("-inf"), dtype=x.dtype, device=x.device).triu_(1)               # clip.py:424-430
    for i in range(clip.txt_layers):
        p = "backbone.transformer.resblocks.%d" % i
        h = layer_norm(x, sd, p + ".ln_1")
        qkv = F.linear(h, sd[p + ".attn.in_proj_weight"], sd[p + ".attn.in_proj_bias"])
        q, k, v = qkv.chunk(3, dim=-1)
        a = mha_core(q, k, v, clip.txt_heads, add_mask=causal)
        x = x + F.linear(a, sd[p + ".attn.out_proj.weight"], sd[p + ".attn.out_proj.bias"])
        h = layer_norm(x, sd, p + ".ln_2")
        h = F.linear(h, sd[p + ".mlp.c_fc.weight"], sd[p + ".mlp.c_fc.bias"])
        h = h * torch.sigmoid(1.702 * h)                                # QuickGELU clip.py:234-236
        x = x + F.linear(h, sd[p + ".mlp.c_proj.weight"], sd[p + ".mlp.c_proj.bias"])
    x = layer_norm(x, sd, "backbone.ln_final")
    eot = word.argmax(dim=-1)                                           # first max wins
    state = x[torch.arange(B, device=x.device), eot] @ sd["backbone.text_projection"]
    if taps is not None:
        taps["word"], taps["state"] = x, state
    return x, state


# ----------------------------------------------------------------------------------------------
# neck / decoder / projector   (reference model/layers.py)
# ----------------------------------------------------------------------------------------------
def conv_bn_relu(x, sd, p, pad, training, bnu):
    # reference model/layers.py:8-11
    return F.relu(batch_norm(F.conv2d(x, sd[p + ".0.weight"], padding=pad), sd, p + ".1", training, bnu))


def fpn(v3, v4, v5, state, sd, training, bnu, taps=None):
    # reference model/layers.py:282-309
    n = "neck"
    s = F.relu(batch_norm(F.linear(state, sd[n + ".txt_proj.0.weight"]), sd, n + ".txt_proj.1", training, bnu))
    f5 = conv_bn_relu(v5, sd, n + ".f1_v_proj", 0, training, bnu)
    f5 = F.relu(batch_norm(f5 * s[:, :, None, None], sd, n + ".norm_layer.0", training, bnu))
    f4 = conv_bn_relu(v4, sd, n + ".f2_v_proj", 1, training, bnu)
    f5_ = F.interpolate(f5, scale_factor=2, mode="bilinear")
    f4 = conv_bn_relu(torch.cat([f4, f5_], 1), sd, n + ".f2_cat", 0, training, bnu)
    f3 = conv_bn_relu(v3, sd, n + ".f3_v_proj", 1, training, bnu)
    f3 = F.avg_pool2d(f3, 2, 2)
    f3 = conv_bn_relu(torch.cat([f3, f4], 1), sd, n + ".f3_cat", 0, training, bnu)
    fq5 = conv_bn_relu(f5, sd, n + ".f4_proj5", 1, training, bnu)
    fq4 = conv_bn_relu(f4, sd, n + ".f4_proj4", 1, training, bnu)
    fq3 = conv_bn_relu(f3, sd, n + ".f4_proj3", 1, training, bnu)
    fq5 = F.interpolate(fq5, scale_factor=2, mode="bilinear")
    fq = conv_bn_relu(torch.cat([fq3, fq4, fq5], 1), sd, n + ".aggr", 0, training, bnu)
    if taps is not None:
        taps["f5"], taps["f4"], taps["f3"], taps["aggr"] = f5, f4, f3, fq
    # CoordConv (layers.py:30-39): x varies along W, y along H, both linspace(-1,1)
    B, _, H, W = fq.shape
    xr = torch.linspace(-1, 1, W).to(fq.dtype).to(fq.device).view(1, 1, 1, W).expand(B, 1, H, W)
    yr = torch.linspace(-1, 1, H).to(fq.dtype).to(fq.device).view(1, 1, H, 1).expand(B, 1, H, W)
    fq = torch.cat([fq, xr, yr], 1)
    fq = conv_bn_relu(fq, sd, n + ".coordconv.0.conv1", 1, training, bnu)
    fq = conv_bn_relu(fq, sd, n + ".coordconv.1", 1, training, bnu)
    return fq


def pos2d(C, H, W):
    # reference model/layers.py:125-152 -> [HW, C]
    pe = torch.zeros(C, H, W)
    d = C // 2
    div = torch.exp(torch.arange(0., d, 2) * -(math.log(10000.0) / d))
    pw = torch.arange(0., W).unsqueeze(1)
    ph = torch.arange(0., H).unsqueeze(1)
    pe[0:d:2] = torch.sin(pw * div).t().unsqueeze(1).repeat(1, H, 1)
    pe[1:d:2] = torch.cos(pw * div).t().unsqueeze(1).repeat(1, H, 1)
    pe[d::2] = torch.sin(ph * div).t().unsqueeze(2).repeat(1, 1, W)
    pe[d + 1::2] = torch.cos(ph * div).t().unsqueeze(2).repeat(1, 1, W)
    return pe.reshape(C, H * W).t().contiguous()


def pos1d(D, L):
    # reference model/layers.py:106-123 -> [L, D]
    pe = torch.zeros(L, D)
    position = torch.arange(0, L).unsqueeze(1)
    div = torch.exp(torch.arange(0, D, 2, dtype=torch.float) * -(math.log(10000.0) / D))
    pe[:, 0::2] = torch.sin(position.float() * div)
    pe[:, 1::2] = torch.cos(position.float() * div)
    return pe


def _mha_module(xq, xk, xv, sd, p, nheads, key_pad, drop, stream):
    E = xq.shape[-1]
    w, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = F.linear(xq, w[:E], b[:E])
    k = F.linear(xk, w[E:2 * E], b[E:2 * E])
    v = F.linear(xv, w[2 * E:], b[2 * E:])
    o = mha_core(q, k, v, nheads, key_pad=key_pad, drop=drop, stream=stream)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def decoder(fq, word, pad_mask, sd, head, drop: DropCtx, taps=None):
    # reference model/layers.py:154-188, 224-250 ; batch-first internally ([B,T,C] == permuted [T,B,C])
    B, C, H, W = fq.shape
    L, D = word.shape[1], word.shape[2]
    vpos = pos2d(C, H, W)[None].to(fq.dtype).to(fq.device)        # [1, HW, C]
    tpos = pos1d(D, L)[None].to(fq.dtype).to(fq.device)           # [1, L, D]
    vis = fq.reshape(B, C, H * W).permute(0, 2, 1)   # [B, HW, C]
    txt = word
    for i in range(head.num_layers):
        p = "decoder.layers.%d" % i
        v2 = layer_norm(vis, sd, p + ".norm1")
        qk = v2 + vpos
        v2 = _mha_module(qk, qk, v2, sd, p + ".self_attn", head.num_head, None, drop, drop_stream(i, 0))
        v2 = layer_norm(v2, sd, p + ".self_attn_norm")
        vis = vis + drop.apply(v2.contiguous(), drop_stream(i, 1))
        v2 = layer_norm(vis, sd, p + ".norm2")
        v2 = _mha_module(v2 + vpos, txt + tpos, txt, sd, p + ".multihead_attn", head.num_head, pad_mask, drop,
                         drop_stream(i, 2))
        v2 = layer_norm(v2, sd, p + ".cross_attn_norm")
        vis = vis + drop.apply(v2.contiguous(), drop_stream(i, 3))
        v2 = layer_norm(vis, sd, p + ".norm3")
        h = F.relu(F.linear(v2, sd[p + ".ffn.0.weight"], sd[p + ".ffn.0.bias"]))
        h = drop.apply(h.contiguous(), drop_stream(i, 4))
        h = layer_norm(h, sd, p + ".ffn.3")
        h = F.linear(h, sd[p + ".ffn.4.weight"], sd[p + ".ffn.4.bias"])
        vis = vis + drop.apply(h.contiguous(), drop_stream(i, 5))
        if taps is not None:
            taps["dec%d" % i] = vis
    out = layer_norm(vis, sd, "decoder.norm")        # [B, HW, C]
    return out.permute(0, 2, 1).reshape(B, C, H, W)


def projector(fq, state, sd, training, bnu):
    # reference model/layers.py:63-84
    p = "proj"
    x = F.interpolate(fq, scale_factor=2, mode="bilinear")
    x = conv_bn_relu(x, sd, p + ".vis.1", 1, training, bnu)
    x = F.interpolate(x, scale_factor=2, mode="bilinear")
    x = conv_bn_relu(x, sd, p + ".vis.3", 1, training, bnu)
    x = F.conv2d(x, sd[p + ".vis.4.weight"], sd[p + ".vis.4.bias"])
    B, C, H, W = x.shape
    wb = F.linear(state, sd[p + ".txt.weight"], sd[p + ".txt.bias"])     # [B, C*9+1]
    weight, bias = wb[:, :-1].reshape(B, C, 3, 3), wb[:, -1]
    out = F.conv2d(x.reshape(1, B * C, H, W), weight, padding=1, groups=B, bias=bias)
    return out.transpose(0, 1)                                            # [B,1,H,W]


def nearest_resize_mask(mask, oh, ow):
    """F.interpolate(mode='nearest') index rule: src = min(floor(dst * in/out), in-1)
    (reference model/segmenter.py:56-58)."""
    ih, iw = mask.shape[-2:]
    ys = torch.clamp((torch.arange(oh, dtype=torch.float32) * (ih / oh)).floor().long(), max=ih - 1).to(mask.device)
    xs = torch.clamp((torch.arange(ow, dtype=torch.float32) * (iw / ow)).floor().long(), max=iw - 1).to(mask.device)
    return mask[..., ys[:, None], xs[None, :]]


def bce_with_logits_mean(x, t):
    # F.binary_cross_entropy_with_logits (segmenter.py:59): max(x,0) - x*t + log1p(exp(-|x|))
    return (x.clamp(min=0) - x * t + torch.log1p(torch.exp(-x.abs()))).mean()


# ----------------------------------------------------------------------------------------------
# whole model   (reference model/segmenter.py:29-62)
# ----------------------------------------------------------------------------------------------
def cris_forward(sd: Dict[str, torch.Tensor], clip, head, img, word, mask=None, training=True,
                 drop_seed: Optional[int] = None, bn_updates: Optional[dict] = None, taps: Optional[dict] = None):
    pad_mask = (word == 0)
    v3, v4, v5 = encode_image(img.to(sd["backbone.visual.conv1.weight"].dtype), sd, clip, training, bn_updates, taps)
    wfeat, state = encode_text(word, sd, clip, taps)
    fq = fpn(v3, v4, v5, state, sd, training, bn_updates, taps)
    if taps is not None:
        taps["fq_neck"] = fq
    drop = DropCtx(head.dropout if training else 0.0, drop_seed if training else None)
    fq = decoder(fq, wfeat, pad_mask, sd, head, drop, taps)
    if taps is not None:
        taps["fq_dec"] = fq
    pred = projector(fq, state, sd, training, bn_updates)
    if not training:
        return pred
    m = mask
    if pred.shape[-2:] != m.shape[-2:]:
        m = nearest_resize_mask(m, pred.shape[-2], pred.shape[-1])
    loss = bce_with_logits_mean(pred, m)
    return pred, m, loss


def train_metric(pred, target, threshold=0.35, pr_iou=0.5):
    """reference utils/misc.py:114-129 (trainMetricGPU)."""
    o = (torch.sigmoid(pred.flatten(1)) >= threshold)
    t = target.flatten(1).bool()
    inter = (o & t).sum(1)
    union = (o | t).sum(1)
    ious = inter / (union + 1e-6)
    return 100.0 * ious.mean(), 100.0 * (ious > pr_iou).float().mean()
