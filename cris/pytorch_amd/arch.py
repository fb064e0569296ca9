"""Architecture description of the CRIS training path + its parameter tree.

The drop-in contract is the reference's *module surface*: parameter /
buffer names and shapes must be those of `model.segmenter.CRIS` (reference
model/segmenter.py:11-27, model/clip.py:147-205,334-386, model/layers.py:47-61,87-104,191-219,
253-280) so `build_segmenter`'s name-prefix grouping, checkpoints, `convert_sync_batchnorm`
and DDP all keep working.  Here the tree is made only of stock torch containers used as
*parameter holders* (nn.Conv2d, nn.BatchNorm2d, nn.Linear, nn.LayerNorm, nn.MultiheadAttention,
nn.Embedding): their `forward` is never called - all arithmetic runs in the HIP library
(csrc/), driven by engine.py.

Also here: deterministic synthetic weights (there is no network for pretrain/RN50.pt), generated
tensor-by-tensor from the parameter *name*, so this container (where the reference can be
imported to make golden fixtures) and the GPU box (where it cannot) build bit-identical models.
"""
from __future__ import annotations

import hashlib
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
from torch import nn


# ----------------------------------------------------------------------------------------------
# specs
# ----------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class ClipSpec:
    """What reference model/clip.py:503-545 infers from a CLIP state_dict."""
    embed_dim: int = 1024
    vision_layers: Tuple[int, int, int, int] = (3, 4, 6, 3)
    vision_width: int = 64
    context_length: int = 77
    vocab_size: int = 49408
    txt_width: int = 512
    txt_layers: int = 12
    pos_grid: int = 7          # attnpool positional_embedding is (pos_grid**2 + 1, width*32)

    @property
    def txt_heads(self):
        return self.txt_width // 64

    @property
    def vis_heads(self):
        return self.vision_width * 32 // 64

    @property
    def vis_embed(self):
        return self.vision_width * 32


@dataclass(frozen=True)
class HeadSpec:
    """Keys CRIS.__init__ reads from the yaml config (reference model/segmenter.py:14-27)."""
    word_len: int = 17
    fpn_in: Tuple[int, int, int] = (512, 1024, 1024)
    fpn_out: Tuple[int, int, int] = (256, 512, 1024)
    num_layers: int = 3
    vis_dim: int = 512
    num_head: int = 8
    dim_ffn: int = 2048
    dropout: float = 0.1
    intermediate: bool = False
    word_dim: int = 1024


CLIP_R50 = ClipSpec()
CLIP_R101 = ClipSpec(embed_dim=512, vision_layers=(3, 4, 23, 3))
HEAD_R50 = HeadSpec()
HEAD_R101 = HeadSpec(fpn_in=(512, 1024, 512), word_dim=512)
# A small member of the same family (same code paths, every channel count still a multiple of 8,
# head dim still 64) for fast CPU/GPU parity tests.
CLIP_TINY = ClipSpec(embed_dim=128, vision_layers=(1, 2, 1, 1), vision_width=16, vocab_size=49408,
                     txt_width=128, txt_layers=2, pos_grid=2)
HEAD_TINY = HeadSpec(word_len=9, fpn_in=(128, 256, 128), fpn_out=(64, 128, 256), num_layers=2,
                     vis_dim=128, num_head=2, dim_ffn=256, dropout=0.1, word_dim=128)


def specs_by_name(name: str):
    name = name.lower()
    if name in ("r50", "cris_r50"):
        return CLIP_R50, HEAD_R50
    if name in ("r101", "cris_r101"):
        return CLIP_R101, HEAD_R101
    if name == "tiny":
        return CLIP_TINY, HEAD_TINY
    raise KeyError(name)


def clip_spec_from_state_dict(sd: Dict[str, torch.Tensor]) -> ClipSpec:
    """Same inference rules as reference model/clip.py:517-542 (ResNet branch only; the ViT
    branch is dead code for every shipped CRIS config - SURVEY.md section 2 row 2)."""
    if "visual.proj" in sd:
        raise NotImplementedError("ViT CLIP archives are outside the CRIS-R50/R101 path")
    counts = tuple(len(set(k.split(".")[2] for k in sd if k.startswith("visual.layer%d" % b)))
                   for b in (1, 2, 3, 4))
    width = sd["visual.layer1.0.conv1.weight"].shape[0]
    npos = sd["visual.attnpool.positional_embedding"].shape[0]
    grid = round((npos - 1) ** 0.5)
    assert grid * grid + 1 == npos
    return ClipSpec(
        embed_dim=sd["text_projection"].shape[1],
        vision_layers=counts, vision_width=width,
        context_length=sd["positional_embedding"].shape[0],
        vocab_size=sd["token_embedding.weight"].shape[0],
        txt_width=sd["ln_final.weight"].shape[0],
        txt_layers=len(set(k.split(".")[2] for k in sd if k.startswith("transformer.resblocks"))),
        pos_grid=grid)


# ----------------------------------------------------------------------------------------------
# parameter tree (holders only - no forward)
# ----------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: arithmetic lives in the HIP engine, not in nn.Module.forward")


def _conv_bn_relu(cin, cout, k=1, pad=0):
    # reference model/layers.py:8-11 (conv_layer)
    return nn.Sequential(nn.Conv2d(cin, cout, k, 1, pad, bias=False), nn.BatchNorm2d(cout), nn.ReLU(True))


class BottleneckP(_Holder):
    # reference model/clip.py:13-42
    def __init__(self, inplanes, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.stride = stride
        self.downsample = None
        if stride > 1 or inplanes != planes * 4:
            self.downsample = nn.Sequential(OrderedDict([
                ("-1", nn.AvgPool2d(stride)),
                ("0", nn.Conv2d(inplanes, planes * 4, 1, stride=1, bias=False)),
                ("1", nn.BatchNorm2d(planes * 4))]))


class AttentionPoolP(_Holder):
    # reference model/clip.py:61-78
    def __init__(self, grid, embed_dim, num_heads, output_dim):
        super().__init__()
        self.spacial_dim = grid
        self.num_heads = num_heads
        self.positional_embedding = nn.Parameter(torch.zeros(grid * grid + 1, embed_dim))
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.c_proj = nn.Linear(embed_dim, output_dim)
        self.connect = nn.Sequential(nn.Conv2d(embed_dim, output_dim, 1, stride=1, bias=False),
                                     nn.BatchNorm2d(output_dim))


class ModifiedResNetP(_Holder):
    # reference model/clip.py:154-205
    def __init__(self, spec: ClipSpec):
        super().__init__()
        w = spec.vision_width
        self.conv1 = nn.Conv2d(3, w // 2, 3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(w // 2)
        self.conv2 = nn.Conv2d(w // 2, w // 2, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(w // 2)
        self.conv3 = nn.Conv2d(w // 2, w, 3, padding=1, bias=False)
        self.bn3 = nn.BatchNorm2d(w)
        inpl = w
        for i, (mult, nblk) in enumerate(zip((1, 2, 4, 8), spec.vision_layers)):
            planes = w * mult
            stride = 1 if i == 0 else 2
            blocks = [BottleneckP(inpl, planes, stride)]
            inpl = planes * 4
            for _ in range(1, nblk):
                blocks.append(BottleneckP(inpl, planes, 1))
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*blocks))
        self.attnpool = AttentionPoolP(spec.pos_grid, w * 32, spec.vis_heads, spec.embed_dim)


class ResidualAttentionBlockP(_Holder):
    # reference model/clip.py:240-253
    def __init__(self, d, heads):
        super().__init__()
        self.attn = nn.MultiheadAttention(d, heads)
        self.ln_1 = nn.LayerNorm(d)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d, d * 4)), ("gelu", nn.Identity()),
                                              ("c_proj", nn.Linear(d * 4, d))]))
        self.ln_2 = nn.LayerNorm(d)


class TextTransformerP(_Holder):
    # reference model/clip.py:268-280
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlockP(width, heads) for _ in range(layers)])


class ClipP(_Holder):
    # reference model/clip.py:334-386 (registration order preserved: state_dict order matters
    # only cosmetically, the key *set* is the contract)
    def __init__(self, spec: ClipSpec):
        super().__init__()
        self.spec = spec
        self.context_length = spec.context_length
        self.visual = ModifiedResNetP(spec)
        self.transformer = TextTransformerP(spec.txt_width, spec.txt_layers, spec.txt_heads)
        self.vocab_size = spec.vocab_size
        self.token_embedding = nn.Embedding(spec.vocab_size, spec.txt_width)
        self.positional_embedding = nn.Parameter(torch.zeros(spec.context_length, spec.txt_width))
        self.ln_final = nn.LayerNorm(spec.txt_width)
        self.text_projection = nn.Parameter(torch.zeros(spec.txt_width, spec.embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))


class CoordConvP(_Holder):
    # reference model/layers.py:19-28
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = _conv_bn_relu(cin + 2, cout, 3, 1)


class FPNP(_Holder):
    # reference model/layers.py:254-280
    def __init__(self, fin, fout):
        super().__init__()
        self.txt_proj = nn.Sequential(nn.Linear(fin[2], fout[2], False), nn.BatchNorm1d(fout[2]), nn.ReLU(True))
        self.f1_v_proj = _conv_bn_relu(fin[2], fout[2], 1, 0)
        self.norm_layer = nn.Sequential(nn.BatchNorm2d(fout[2]), nn.ReLU(True))
        self.f2_v_proj = _conv_bn_relu(fin[1], fout[1], 3, 1)
        self.f2_cat = _conv_bn_relu(fout[2] + fout[1], fout[1], 1, 0)
        self.f3_v_proj = _conv_bn_relu(fin[0], fout[0], 3, 1)
        self.f3_cat = _conv_bn_relu(fout[0] + fout[1], fout[1], 1, 0)
        self.f4_proj5 = _conv_bn_relu(fout[2], fout[1], 3, 1)
        self.f4_proj4 = _conv_bn_relu(fout[1], fout[1], 3, 1)
        self.f4_proj3 = _conv_bn_relu(fout[1], fout[1], 3, 1)
        self.aggr = _conv_bn_relu(3 * fout[1], fout[1], 1, 0)
        self.coordconv = nn.Sequential(CoordConvP(fout[1], fout[1]), _conv_bn_relu(fout[1], fout[1], 3, 1))


class DecoderLayerP(_Holder):
    # reference model/layers.py:192-219
    def __init__(self, d, nhead, dff, dropout):
        super().__init__()
        self.self_attn_norm = nn.LayerNorm(d)
        self.cross_attn_norm = nn.LayerNorm(d)
        self.self_attn = nn.MultiheadAttention(d, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d, nhead, dropout=dropout, kdim=d, vdim=d)
        self.ffn = nn.Sequential(nn.Linear(d, dff), nn.ReLU(True), nn.Dropout(dropout), nn.LayerNorm(dff),
                                 nn.Linear(dff, d))
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)
        self.norm3 = nn.LayerNorm(d)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)


class DecoderP(_Holder):
    # reference model/layers.py:88-104
    def __init__(self, head: HeadSpec):
        super().__init__()
        self.layers = nn.ModuleList([DecoderLayerP(head.vis_dim, head.num_head, head.dim_ffn, head.dropout)
                                     for _ in range(head.num_layers)])
        self.num_layers = head.num_layers
        self.norm = nn.LayerNorm(head.vis_dim)
        self.return_intermediate = head.intermediate


class ProjectorP(_Holder):
    # reference model/layers.py:48-61
    def __init__(self, word_dim, in_dim, k):
        super().__init__()
        self.in_dim, self.kernel_size = in_dim, k
        self.vis = nn.Sequential(nn.Upsample(scale_factor=2, mode="bilinear"),
                                 _conv_bn_relu(in_dim * 2, in_dim * 2, 3, 1),
                                 nn.Upsample(scale_factor=2, mode="bilinear"),
                                 _conv_bn_relu(in_dim * 2, in_dim, 3, 1),
                                 nn.Conv2d(in_dim, in_dim, 1))
        self.txt = nn.Linear(word_dim, in_dim * k * k + 1)


def build_param_tree(clip: ClipSpec, head: HeadSpec) -> nn.Module:
    """A bare nn.Module with children backbone / neck / decoder / proj (reference
    model/segmenter.py:16-27).  The drop-in CRIS module (model/segmenter.py here) subclasses
    nn.Module and attaches these four children under the same names."""
    root = nn.Module()
    root.backbone = ClipP(clip)
    root.neck = FPNP(list(head.fpn_in), list(head.fpn_out))
    root.decoder = DecoderP(head)
    root.proj = ProjectorP(head.word_dim, head.vis_dim // 2, 3)
    return root


# ----------------------------------------------------------------------------------------------
# deterministic synthetic weights
# ----------------------------------------------------------------------------------------------
# The pixel-text similarity (the logits) sums 2304 products x*w with w = Linear(state): at plain fan-in scales its std is
# ~50.  The shrink needed for O(1) logits (initial BCE ~0.8) is spread over the three tensors on that product path
# (text_projection, proj.txt.weight, proj.vis.4.weight) so that none of them becomes tiny relative to Adam's fixed
# 1e-4 first steps (a 100x smaller weight is 100x more sensitive: the loss then jumps 0.8 -> 11 after ONE step).
LOGIT_SHRINK = 0.25


def _gen_for(name: str, seed: int) -> torch.Generator:
    h = hashlib.sha256(("%d:%s" % (seed, name)).encode()).digest()
    g = torch.Generator(device="cpu")
    g.manual_seed(int.from_bytes(h[:8], "little") & 0x7FFFFFFFFFFFFFFF)
    return g


def _round_fp16(t: torch.Tensor) -> torch.Tensor:
    return t.half().float()


def synthetic_state_dict(clip: ClipSpec, head: HeadSpec, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Full CRIS state_dict (reference key set) with seeded random values.

    Scale rules keep activations O(1) through the net (He-style fan-in for convs/linears, BN gammas
    around 1, small gammas on each bottleneck's last BN).  CLIP conv / linear / attention weights are
    rounded to fp16-representable values, mirroring what the reference's loader does to a real CLIP
    archive (model/clip.py:477-500,552 then model/segmenter.py:16 `.float()`), so a state_dict
    produced here survives the reference's `build_model` unchanged.
    """
    tree = build_param_tree(clip, head)
    out = OrderedDict()
    for name, ref in tree.state_dict().items():
        g = _gen_for(name, seed)
        shape = tuple(ref.shape)
        leaf = name.rsplit(".", 1)[-1]
        parent = name.rsplit(".", 1)[0] if "." in name else ""
        in_clip = name.startswith("backbone.")
        if leaf == "num_batches_tracked":
            t = torch.zeros(shape, dtype=torch.int64)
        elif leaf == "running_mean":
            t = torch.randn(shape, generator=g) * 0.1
        elif leaf == "running_var":
            t = torch.rand(shape, generator=g) + 0.5
        elif name == "backbone.logit_scale":
            t = torch.ones(shape) * math.log(1 / 0.07)
        elif name == "backbone.token_embedding.weight":
            t = torch.randn(shape, generator=g) * 0.02
        elif name == "backbone.positional_embedding":
            t = torch.randn(shape, generator=g) * 0.01
        elif name.endswith("attnpool.positional_embedding"):
            t = torch.randn(shape, generator=g) / shape[1] ** 0.5
        elif name == "backbone.text_projection":
            t = _round_fp16(torch.randn(shape, generator=g) * shape[0] ** -0.5 * LOGIT_SHRINK)
        elif len(shape) == 1 and leaf in ("weight", "bias") and _is_norm(tree, parent):
            if leaf == "weight":
                t = _norm_gamma(parent, shape, seed)
            else:
                t = torch.randn(shape, generator=g) * 0.05
                if _is_batchnorm(tree, parent):
                    # beta = +gamma: ~84 % of the ReLU units behind each BatchNorm are active, like a trained network and
                    # unlike a random ReLU-BN stack, whose forward map is chaotic (a single bf16 rounding per layer grows
                    # to 10-50 % at the logits: measured with oracle/bf16_emulation.py).  Any state_dict is a legitimate
                    # test input; this one keeps bf16-vs-fp32 parity figures meaningful.
                    t = t + _norm_gamma(parent, shape, seed)
        elif leaf in ("bias", "in_proj_bias"):
            t = torch.randn(shape, generator=g) * 0.02
            if name.startswith("proj.txt."):
                t = t * 0.5         # 2304 of these enter every logit (see the weight rule below)
            if in_clip:
                t = _round_fp16(t)
        elif leaf in ("weight", "in_proj_weight"):
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            gain = 2.0 if len(shape) == 4 else 1.0
            t = torch.randn(shape, generator=g) * math.sqrt(gain / fan_in)
            if name in ("proj.txt.weight", "proj.vis.4.weight"):
                t = t * LOGIT_SHRINK
            if in_clip:
                t = _round_fp16(t)
        else:  # pragma: no cover
            raise KeyError("no init rule for %s %s" % (name, shape))
        out[name] = t.contiguous()
    return out


def _norm_gamma(parent: str, shape, seed: int) -> torch.Tensor:
    lo, hi = (0.2, 0.6) if parent.endswith(".bn3") else (0.7, 1.3)
    return torch.rand(shape, generator=_gen_for(parent + ".weight", seed)) * (hi - lo) + lo


def _is_batchnorm(tree: nn.Module, parent: str) -> bool:
    return isinstance(tree.get_submodule(parent), nn.modules.batchnorm._BatchNorm)


def _is_norm(tree: nn.Module, parent: str) -> bool:
    m = tree.get_submodule(parent)
    return isinstance(m, (nn.modules.batchnorm._BatchNorm, nn.LayerNorm))


def clip_state_dict_view(full_sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The `backbone.*` part with the prefix stripped = what a CLIP archive's state_dict holds
    (plus the 3 scalar keys the reference deletes, model/clip.py:548-550)."""
    return OrderedDict((k[len("backbone."):], v) for k, v in full_sd.items() if k.startswith("backbone."))
