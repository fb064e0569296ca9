"""Stage-isolated forward+backward parity (diagnostic): every stage gets the oracle's bf16-rounded inputs and a random
upstream gradient; input- and parameter-gradients are compared with torch autograd over the oracle's stage function."""
import dataclasses
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import arch, synth  # noqa: E402
from cris.pytorch_amd.engine import Engine, Act  # noqa: E402
from oracle import cris_oracle as O  # noqa: E402
from tools.parity_report import rel, cos, nhwc_to_nchw  # noqa: E402

BF = torch.bfloat16
DEV = "cuda"


def r16(x):
    return x.detach().to(BF).float()


def to_act(x):
    B, C, H, W = x.shape
    return Act(x.permute(0, 2, 3, 1).reshape(B * H * W, C).to(DEV).to(BF).contiguous(), B, H, W, C)


def set_grad(act, g_nchw):
    B, C, H, W = g_nchw.shape
    act.root._g = g_nchw.permute(0, 2, 3, 1).reshape(B * H * W, C).to(DEV).to(BF).contiguous()


def grad_nchw(act):
    g = act.g[:, act.coff:act.coff + act.C].float()
    return g.view(act.Bn, act.H, act.W, act.C).permute(0, 3, 1, 2)


def rand_like(x, seed):
    return r16(torch.randn(x.shape, generator=torch.Generator().manual_seed(seed)))


RESULTS = []          # (stage, what, rel, cos) rows collected by record()


def record(stage, what, a, b):
    r, c = rel(a, b), cos(a, b)
    RESULTS.append((stage, what, r, c))
    return r, c


class Ctx:
    def __init__(self, spec, B, S, dropout=0.0):
        self.clip, head = arch.specs_by_name(spec)
        self.head = dataclasses.replace(head, dropout=dropout)
        self.sd = arch.synthetic_state_dict(self.clip, self.head, 0)
        self.img, self.word, self.mask = synth.make_batch(B, S, self.head.word_len, 0, 0)
        params = {k: v.to(DEV) for k, v in self.sd.items() if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
        buffers = {k: v.to(DEV) for k, v in self.sd.items() if k.endswith(("running_mean", "running_var"))}
        self.eng = Engine(self.clip, self.head, params, buffers, DEV)
        self.taps = {}
        with torch.no_grad():
            O.cris_forward(self.sd, self.clip, self.head, self.img, self.word, self.mask, training=True, taps=self.taps)

    def begin(self, seed=5):
        e = self.eng
        e.training, e.seed, e.tape, e._dgrad_outT = True, seed, [], None
        Act._engine = None
        e.grad_arena.zero_()
        e.repack_weights()
        self.leaf = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in self.sd.items()}

    def finish(self, name, pairs, prefix):
        e = self.eng
        for fn in reversed(e.tape):
            fn()
        e._flush_queues()                    # queued weight gradients / LayerNorm parameter-gradient sums
        e.tape = []
        torch.cuda.synchronize()
        out = []
        for what, a, b in pairs:
            out.append("%s: rel %.2e cos %.5f" % ((what,) + record(name, what, a, b)))
        worst = (2.0, 0.0, "")
        n = 0
        for k, v in self.leaf.items():
            if k.startswith(prefix) and v.is_floating_point() and v.grad is not None and not k.endswith("k_proj.bias"):
                c = cos(e.grad_param_layout(k), v.grad)
                RESULTS.append((name, "param:" + k, rel(e.grad_param_layout(k), v.grad), c))
                n += 1
                if c < worst[0]:
                    worst = (c, rel(e.grad_param_layout(k), v.grad), k)
        print("[%s] %s | params %d worst cos %.5f (rel %.2e) %s" % (name, " ; ".join(out), n, worst[0], worst[1], worst[2]))
        if os.environ.get("VERBOSE_STAGE") == name:
            for k, v in self.leaf.items():
                if k.startswith(prefix) and v.is_floating_point() and v.grad is not None:
                    print("        %.5f %.2e %s" % (cos(e.grad_param_layout(k), v.grad), rel(e.grad_param_layout(k), v.grad), k))


def main(spec="tiny", B=4, S=64, blocks=("layer2.0", "layer2.1")):
    """blocks: the bottlenecks to check ("layerL.K"); block K > 0 gets the oracle's own output of blocks 0 .. K-1 as its input"""
    del RESULTS[:]
    c = Ctx(spec, B, S)
    e, t = c.eng, c.taps
    v = "backbone.visual"
    # ---- bottlenecks
    w = e.clip.vision_width
    for blk in blocks:
        li, bi = int(blk[5]), int(blk.split(".")[1])
        planes = w * (1, 2, 4, 8)[li - 1]
        c.begin()
        x0 = t["layer%d" % (li - 1)] if li > 1 else t["stem"]
        with torch.no_grad():
            for k in range(bi):                      # the oracle's own chain up to the block under test
                x0 = O.bottleneck(r16(x0), c.sd, "%s.layer%d.%d" % (v, li, k), 2 if (li > 1 and k == 0) else 1, True, None)
        stride = 2 if (li > 1 and bi == 0) else 1
        has_ds = bi == 0
        xl = r16(x0).requires_grad_(True)
        ref = O.bottleneck(xl, c.leaf, "%s.%s" % (v, blk), stride, True, None)
        gout = rand_like(ref, 1)
        ref.backward(gout)
        xa = to_act(r16(x0))
        z = e._bottleneck(xa, "%s.%s" % (v, blk), planes, stride, has_ds)
        set_grad(z, gout)
        c.finish("bottleneck " + blk, [("out", nhwc_to_nchw(z), ref)], "%s.%s" % (v, blk))
        print("      dx: rel %.2e cos %.5f" % record("bottleneck " + blk, "dx", grad_nchw(xa), xl.grad))
    # ---- attnpool
    c.begin()
    xl = r16(t["layer4"]).requires_grad_(True)
    ref = O.attnpool(xl, c.leaf, v + ".attnpool", e.clip.vis_heads, e.clip.pos_grid, True, None)
    gout = rand_like(ref, 2)
    ref.backward(gout)
    xa = to_act(r16(t["layer4"]))
    z = e._attnpool(xa, v + ".attnpool")
    set_grad(z, gout)
    c.finish("attnpool", [("out", nhwc_to_nchw(z), ref)], v + ".attnpool")
    print("      dx: rel %.2e cos %.5f" % record("attnpool", "dx", grad_nchw(xa), xl.grad))
    # ---- text encoder
    c.begin()
    wref, sref = O.encode_text(c.word, c.leaf, e.clip)
    gw, gs = rand_like(wref, 3), rand_like(sref, 4)
    (wref * gw).sum().backward(retain_graph=True)
    (sref * gs).sum().backward()
    e._text_tape_start = len(e.tape)
    xf, state = e._encode_text(c.word.to(DEV))
    xf.root._g = gw.reshape(-1, gw.shape[-1]).to(DEV).to(BF).contiguous()
    state.root._g = gs.to(DEV).to(state.t.dtype).contiguous()             # (fp32 when the sentence vector stays in fp32)
    c.finish("text", [("word", xf.t.float().view(wref.shape), wref), ("state", state.t.float(), sref)], "backbone.t")
    print("      token_embedding cos %.5f pos cos %.5f text_projection cos %.5f ln_final.w cos %.5f" % tuple(
        cos(e.grad_param_layout(k), c.leaf[k].grad) for k in ("backbone.token_embedding.weight", "backbone.positional_embedding",
                                              "backbone.text_projection", "backbone.ln_final.weight")))
    # ---- FPN
    c.begin()
    ins = [r16(t["layer2"]).requires_grad_(True), r16(t["layer3"]).requires_grad_(True), r16(t["attnpool"]).requires_grad_(True),
           r16(t["state"]).requires_grad_(True)]
    ref = O.fpn(ins[0], ins[1], ins[2], ins[3], c.leaf, True, None)
    gout = rand_like(ref, 5)
    ref.backward(gout)
    acts = [to_act(x.detach()) for x in ins[:3]]
    sdt = torch.float32 if (e.state_f32 and B <= 16) else BF
    st = Act(ins[3].detach().to(DEV).to(sdt).contiguous(), B, 1, 1, ins[3].shape[1])
    z = e._fpn(acts[0], acts[1], acts[2], st)
    set_grad(z, gout)
    c.finish("fpn", [("out", nhwc_to_nchw(z), ref)], "neck")
    for nm, a, l in zip(("dv3", "dv4", "dv5"), acts, ins):
        print("      %s: rel %.2e cos %.5f" % ((nm,) + record("fpn", nm, grad_nchw(a), l.grad)))
    print("      dstate: rel %.2e cos %.5f" % record("fpn", "dstate", st.g.float(), ins[3].grad))
    # ---- decoder (with dropout masks from the shared hash)
    for dp in (0.0, 0.1):
        c2 = Ctx(spec, B, S, dropout=dp) if dp > 0 else c
        e2, t2 = c2.eng, c2.taps
        c2.begin(seed=77)
        fql = r16(t2["fq_neck"]).requires_grad_(True)
        wl = r16(t2["word"]).requires_grad_(True)
        drop = O.DropCtx(dp, 77 if dp > 0 else None)
        ref = O.decoder(fql, wl, c2.word == 0, c2.leaf, c2.head, drop)
        gout = rand_like(ref, 6)
        ref.backward(gout)
        fa = to_act(fql.detach())
        ta = Act(wl.detach().reshape(-1, wl.shape[-1]).to(DEV).to(BF).contiguous(), B, wl.shape[1], 1, wl.shape[-1])
        e2._dec_tape_start = len(e2.tape)
        z = e2._decoder(fa, ta, c2.word.to(DEV))
        set_grad(z, gout)
        c2.finish("decoder p=%g" % dp, [("out", nhwc_to_nchw(z), ref)], "decoder")
        print("      dfq: rel %.2e cos %.5f ; dtxt: rel %.2e cos %.5f" % (record("decoder p=%g" % dp, "dfq", grad_nchw(fa), fql.grad)
                                                                      + record("decoder p=%g" % dp, "dtxt", ta.g.float().view(wl.shape), wl.grad)))
    # ---- projector + loss
    c.begin()
    fql = r16(t["fq_dec"]).requires_grad_(True)
    sl = r16(t["state"]).requires_grad_(True)
    pref = O.projector(fql, sl, c.leaf, True, None)
    m = O.nearest_resize_mask(c.mask, pref.shape[-2], pref.shape[-1])
    lref = O.bce_with_logits_mean(pref, m)
    lref.backward()
    fa = to_act(fql.detach())
    st = Act(sl.detach().to(DEV).to(sdt).contiguous(), B, 1, 1, sl.shape[1])
    pred, x, wb = e._projector(fa, st)
    OH, OW = pred.shape[-2:]
    from cris.pytorch_amd import ops
    msk = torch.empty(B, 1, OH, OW, device=DEV)
    ops.mask_resize_nearest(c.mask.to(DEV), OH, OW, msk)
    loss = torch.zeros(1, device=DEV)
    ops.bce_fwd(pred, msk, loss)
    cc = e.head.vis_dim // 2

    def bwd_loss():
        dpred = torch.empty(B, 1, OH, OW, device=DEV)
        ops.bce_bwd(pred, msk, None, dpred)
        gx, acc = x.grad_target()
        e._dwb = torch.zeros(B, wb.ld, device=DEV)
        ops.dynconv_bwd(x.t, dpred, B, OH, OW, cc, wb.t, gx, e._dwb)
    e.tape.append(bwd_loss)
    c.finish("projector+loss", [("pred", pred, pref), ("loss", loss, lref.detach().view(1))], "proj")
    print("      dfq: rel %.2e cos %.5f ; dstate: rel %.2e cos %.5f" % (record("projector+loss", "dfq", grad_nchw(fa), fql.grad)
                                                                        + record("projector+loss", "dstate", st.g.float(), sl.grad)))
    return RESULTS


if __name__ == "__main__":
    main(*(sys.argv[1:2] or ["tiny"]), *[int(a) for a in sys.argv[2:4]], **({"blocks": tuple(sys.argv[4].split(","))} if len(sys.argv) > 4 else {}))
