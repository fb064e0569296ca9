"""TEST INFRASTRUCTURE ONLY - run the fp32 oracle with bf16 *storage* rounding at the points where the HIP path
stores bf16 (conv / linear inputs, weights and outputs, ReLU / BN-apply outputs, LayerNorm outputs, attention
probabilities and outputs, resampling outputs), fp32 everywhere else (accumulation, statistics, residual streams).

Purpose: separate implementation errors from the inherent bf16 noise of this network.  With the synthetic
random weights a 50-layer ResNet amplifies one bf16 rounding per layer to ~13 % relative error at layer4 and
~30 % after the neck *for any bf16 implementation*; the HIP path must match this emulation closely (same
rounding points => mostly identical rounded values) and the fp32 oracle only to that noise floor.

`bf16_storage(kinds=..., where=...)` restricts the rounding to some KINDS of storage point and to the part of the
network for which `where()` is true - tools/error_budget.py switches the points on one group at a time to find which
of them the loss error of the bf16 path comes from.
"""
import contextlib

import torch

from . import cris_oracle as O

BF = torch.bfloat16

# kinds of storage point (what gets rounded)
KINDS = ("conv_in", "conv_w", "conv_out", "linear_in", "linear_w", "linear_out", "relu", "interp", "pool", "ln", "attn_p", "attn_out")


def r(x):
    return x.to(BF).to(x.dtype)


# Which part of the network a call belongs to: the oracle's stage functions are wrapped so that STAGE[0] names the stage
# being computed (stem, layer1..4, attnpool, text, neck, decoder, proj) - see staged().
STAGE = [None]
STAGES = ("stem", "layer1", "layer2", "layer3", "layer4", "attnpool", "text", "neck", "decoder", "proj")


@contextlib.contextmanager
def staged():
    """wrap the oracle's stage functions so that STAGE[0] is set while they run (forward only - the backward of the rounding
    is the identity, so nothing else needs to know the stage)"""
    saved = dict(bottleneck=O.bottleneck, attnpool=O.attnpool, encode_text=O.encode_text, fpn=O.fpn, decoder=O.decoder,
                 projector=O.projector, encode_image=O.encode_image)

    def scoped(name_of, fn):
        def run(*a, **k):
            prev = STAGE[0]
            STAGE[0] = name_of(*a, **k)
            try:
                return fn(*a, **k)
            finally:
                STAGE[0] = prev
        return run

    O.encode_image = scoped(lambda *a, **k: "stem", saved["encode_image"])          # until a bottleneck / attnpool takes over
    O.bottleneck = scoped(lambda x, sd, p, *a, **k: p.split(".")[2], saved["bottleneck"])
    O.attnpool = scoped(lambda *a, **k: "attnpool", saved["attnpool"])
    O.encode_text = scoped(lambda *a, **k: "text", saved["encode_text"])
    O.fpn = scoped(lambda *a, **k: "neck", saved["fpn"])
    O.decoder = scoped(lambda *a, **k: "decoder", saved["decoder"])
    O.projector = scoped(lambda *a, **k: "proj", saved["projector"])
    try:
        yield
    finally:
        for k, v in saved.items():
            setattr(O, k, v)
        STAGE[0] = None


@contextlib.contextmanager
def bf16_storage(kinds=None, where=None, skip=None):
    """kinds: the storage points to round (None: all of KINDS); where: optional predicate, evaluated at every storage point -
    rounding happens only while it is true (e.g. `lambda: STAGE[0] == "decoder"` under staged()); skip: optional predicate of
    the point's kind - a point it accepts is left in fp32 (e.g. `lambda kind: STAGE[0] == "text" and kind == "linear_w"`:
    everything rounded except the text encoder's weights - what promoting one group of points would buy)"""
    kinds = set(KINDS if kinds is None else kinds)
    assert kinds <= set(KINDS), kinds - set(KINDS)
    saved = dict(conv2d=O.F.conv2d, linear=O.F.linear, relu=O.F.relu, interpolate=O.F.interpolate, avg_pool2d=O.F.avg_pool2d,
                 layer_norm=O.layer_norm, mha_core=O.mha_core)

    def rk(x, kind):
        if kind in kinds and (where is None or where()) and not (skip is not None and skip(kind)):
            return r(x)
        return x

    def conv2d(x, w, b=None, **kw):
        b = kw.pop("bias", b)
        return rk(saved["conv2d"](rk(x, "conv_in"), rk(w, "conv_w"), b, **kw), "conv_out")

    def linear(x, w, b=None):
        return rk(saved["linear"](rk(x, "linear_in"), rk(w, "linear_w"), b), "linear_out")

    def relu(x, *a, **k):
        return rk(saved["relu"](x), "relu")

    def interpolate(x, *a, **k):
        return rk(saved["interpolate"](x, *a, **k), "interp")

    def avg_pool2d(x, *a, **k):
        return rk(saved["avg_pool2d"](x, *a, **k), "pool")

    def layer_norm(x, sd, prefix):
        return rk(saved["layer_norm"](x, sd, prefix), "ln")

    def mha_core(q, k, v, nheads, add_mask=None, key_pad=None, drop=None, stream=0):
        B, Lq, E = q.shape
        Lk = k.shape[1]
        d = E // nheads
        qh = q.view(B, Lq, nheads, d).transpose(1, 2)
        kh = k.view(B, Lk, nheads, d).transpose(1, 2)
        vh = v.view(B, Lk, nheads, d).transpose(1, 2)
        s = (qh @ kh.transpose(-1, -2)) * (d ** -0.5)
        if add_mask is not None:
            s = s + add_mask
        if key_pad is not None:
            s = s.masked_fill(key_pad[:, None, None, :], float("-inf"))
        m = s.max(-1, keepdim=True).values
        e = torch.exp(s - m)
        l = e.sum(-1, keepdim=True)
        p = e
        if drop is not None and drop.active:
            p = drop.apply(p.contiguous(), stream)
        o = (rk(p, "attn_p") @ vh) / l                # probabilities enter the PV product in bf16, normalised after
        return rk(o.transpose(1, 2).reshape(B, Lq, E), "attn_out")

    O.F.conv2d, O.F.linear, O.F.relu, O.F.interpolate, O.F.avg_pool2d = conv2d, linear, relu, interpolate, avg_pool2d
    O.layer_norm, O.mha_core = layer_norm, mha_core
    try:
        yield
    finally:
        O.F.conv2d, O.F.linear, O.F.relu = saved["conv2d"], saved["linear"], saved["relu"]
        O.F.interpolate, O.F.avg_pool2d = saved["interpolate"], saved["avg_pool2d"]
        O.layer_norm, O.mha_core = saved["layer_norm"], saved["mha_core"]
