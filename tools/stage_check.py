"""Stage-isolated parity: feed each engine stage the ORACLE's inputs (bf16-rounded) and compare its outputs with
the oracle's, so per-stage bugs are separated from error amplification through the network.  Diagnostic."""
import dataclasses
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import arch, synth  # noqa: E402
from cris.pytorch_amd.engine import Engine, Act  # noqa: E402
from oracle import cris_oracle as O  # noqa: E402
from tools.parity_report import rel, nhwc_to_nchw  # noqa: E402

BF = torch.bfloat16


def to_act(x, dev):
    B, C, H, W = x.shape
    t = x.permute(0, 2, 3, 1).reshape(B * H * W, C).to(dev).to(BF).contiguous()
    return Act(t, B, H, W, C)


def main(spec="tiny", B=4, S=64):
    clip, head = arch.specs_by_name(spec)
    head = dataclasses.replace(head, dropout=0.0)
    sd = arch.synthetic_state_dict(clip, head, 0)
    img, word, mask = synth.make_batch(B, S, head.word_len, 0, 0)
    dev = "cuda"
    params = {k: v.to(dev) for k, v in sd.items() if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
    buffers = {k: v.to(dev) for k, v in sd.items() if k.endswith(("running_mean", "running_var"))}
    eng = Engine(clip, head, params, buffers, dev)
    ot = {}
    with torch.no_grad():
        O.cris_forward(sd, clip, head, img, word, mask, training=True, taps=ot)
    eng.training, eng.seed, eng.tape, eng._dgrad_outT = True, 0, [], None
    Act._engine = None
    eng.repack_weights()
    # determinism of the full forward
    t1, t2 = {}, {}
    eng.forward(img.to(dev), word.to(dev), mask.to(dev), True, 0, t1)
    p1 = {k: (v.t.clone() if isinstance(v, Act) else v.clone()) for k, v in t1.items()}
    eng.forward(img.to(dev), word.to(dev), mask.to(dev), True, 0, t2)
    for k in ("layer1", "layer2", "layer3", "layer4", "attnpool", "word", "state", "s", "f5", "f4", "f3", "fq_neck", "fq_dec", "pred"):
        a = p1[k]
        b = t2[k].t if isinstance(t2[k], Act) else t2[k]
        print("determinism %-9s max|d| %.3e  rel %.3e" % (k, float((a.float() - b.float()).abs().max()), rel(a, b)))
    eng.tape = []
    # --- FPN fed with oracle inputs
    x2, x3 = ot["layer2"], ot["layer3"]
    v3, v4, v5 = to_act(x2, dev), to_act(x3, dev), to_act(ot["attnpool"], dev)
    st = ot["state"]
    state = Act(st.to(dev).to(torch.float32 if (eng.state_f32 and B <= 16) else BF).contiguous(), B, 1, 1, st.shape[1])
    fq = eng._fpn(v3, v4, v5, state)
    nt = eng._neck_taps
    for k in ("f5", "f4", "f3", "aggr"):
        print("FPN(oracle in) %-6s rel %.3e" % (k, rel(nhwc_to_nchw(nt[k]), ot[k])))
    print("FPN(oracle in) fq     rel %.3e" % rel(nhwc_to_nchw(fq), ot["fq_neck"]))
    # --- decoder fed with oracle inputs
    wfeat = ot["word"]
    txt = Act(wfeat.reshape(-1, wfeat.shape[-1]).to(dev).to(BF).contiguous(), B, wfeat.shape[1], 1, wfeat.shape[-1])
    eng._dec_tape_start = len(eng.tape)
    fqd = eng._decoder(to_act(ot["fq_neck"], dev), txt, word.to(dev))
    print("DEC(oracle in) fq_dec rel %.3e" % rel(nhwc_to_nchw(fqd), ot["fq_dec"]))
    for i in range(head.num_layers):
        pass
    # --- projector fed with oracle inputs
    pred, x, wb = eng._projector(to_act(ot["fq_dec"], dev), state)
    with torch.no_grad():
        opred = O.projector(ot["fq_dec"], ot["state"], sd, True, None)
    print("PROJ(oracle in) pred  rel %.3e" % rel(pred, opred))
    # --- visual stages
    v3b, v4b, v5b, feats = eng._encode_image(img.to(dev))
    for i, k in enumerate(("layer1", "layer2", "layer3", "layer4")):
        print("VIS  %-8s rel %.3e" % (k, rel(nhwc_to_nchw(feats[i]), ot[k])))
    ap = eng._attnpool(to_act(ot["layer4"], dev), "backbone.visual.attnpool")
    print("ATTNPOOL(oracle in) rel %.3e" % rel(nhwc_to_nchw(ap), ot["attnpool"]))


if __name__ == "__main__":
    main(*(sys.argv[1:2] or ["tiny"]), *[int(a) for a in sys.argv[2:4]])
