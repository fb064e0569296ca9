"""Inference path (reference engine/engine.py:90-123,171-188, test.py:70-86, tools/latency.py:38-72):
the eval-mode forward with every BatchNorm FOLDED into its convolution, replayed as one HIP graph.

`model.eval()` BatchNorm is a per-channel affine map with constants (running statistics), so
    relu(bn(conv_W(x)))            = relu(conv_{W*s}(x) + t)               s = gamma / sqrt(var + eps), t = beta - mean * s
    relu(bn3(conv3(a)) + identity) = relu(conv_{W3*s3}(a) + t3 + identity)  (Bottleneck tail, model/clip.py:52-56)
    relu(bn3(conv3(a)) + bnd(convd(xi)))                                    (downsample branch: two folded GEMMs, the second adds the first)
The scaled weights are packed once into their bf16 GEMM layout (cris_pack_weights with `row_scale`), the shift becomes the
GEMM's bias, ReLU (before or after the residual add) its activation: 69 of the 71 BatchNorm layers cost no launch and no pass
over the activations at all (the reference runs native_batch_norm + relu as separate kernels in eval mode too).  Not folded:
`neck.f1_v_proj` (its output is multiplied by the sentence vector before the next BatchNorm) and `neck.norm_layer`.

`InferEngine` reuses the training engine's schedule (engine.Engine) - the network is described once; the folding hooks in at
the two layer primitives (`gemm(stats=True)` defers its launch, `bn()` issues it fused).  `InferenceRunner` is what
`engine.validate` / `inference` / tools/latency.py need: static input buffers, the forward (and optionally the evaluation
post-processing: sigmoid + bicubic upsampling to the input size, evalpost.py) captured into ONE HIP graph per input shape.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops
from .arch import ClipSpec, HeadSpec
from .engine import Act, BN_EPS, Engine
from .ops import Geom, pad8

F32 = torch.float32


class _Pending:
    """A convolution whose launch is deferred until the BatchNorm that follows it is known (InferEngine.gemm / .bn)."""
    __slots__ = ("x", "wname", "N", "g")

    def __init__(self, x: Act, wname: str, N: int, g: Geom):
        self.x, self.wname, self.N, self.g = x, wname, N, g


class InferEngine(Engine):
    def __init__(self, clip: ClipSpec, head: HeadSpec, params: Dict[str, torch.Tensor], buffers: Dict[str, torch.Tensor], device,
                 fold_bn: bool = True):
        super().__init__(clip, head, params, buffers, device, inference_only=True)
        self.fold_bn = fold_bn
        self._fold = {}                 # (conv weight name, BatchNorm prefix) -> (bf16 F pack of W * s, shift t, keep-alive)

    def forward(self, img, word, mask=None, training=False, **kw):
        if training:
            raise RuntimeError("InferEngine is built without a gradient arena: use Engine / NativeTrainer for training")
        out = super().forward(img, word, mask, training=training, **kw)
        if not training:
            self.packs_current = True       # frozen weights: the bf16 operand copies made by this forward stay valid
            self._zneed_last = max(self._zneed_last, self._zneed)      # size of the zero slab the next forward carves from
        return out

    def invalidate(self):
        """parameters or running statistics changed (load_state_dict): fold and pack again on the next forward"""
        self._fold = {}
        self.packs_current = False

    # -- folded operands ---------------------------------------------------------------------------------------------
    def _folded(self, wname: str, pfx: str):
        key = (wname, pfx)
        hit = self._fold.get(key)
        if hit is not None:
            return hit[0], hit[1]
        assert not torch.cuda.is_current_stream_capturing(), "folded weights must exist before the graph capture (run one eager forward)"
        w = self.P[wname]
        lay = self.gemm_layout(wname)
        N, Cin, taps, Cpad = lay if lay is not None else (w.shape[0], w.shape[1], 1, pad8(w.shape[1]))
        scale = torch.empty(N, dtype=F32, device=self.dev)
        shift = torch.empty(N, dtype=F32, device=self.dev)
        ops.bn_eval_coeffs(self.P[pfx + ".weight"], self.P[pfx + ".bias"], self.Bf[pfx + ".running_mean"],
                           self.Bf[pfx + ".running_var"], BN_EPS, N, scale, shift)
        tab = ops.PackTable()
        wf, _ = tab.add(w, N, Cin, taps, Cpad=Cpad, want_D=False, row_scale=scale)
        tab.run()
        self._fold[key] = (wf, shift, tab, scale)
        return wf, shift

    def _launch(self, pd: _Pending, wf, bias, act: int, resid: Optional[Act], out: Optional[Act]) -> Act:
        g, x = pd.g, pd.x
        if out is None:
            out = self.new_act(g.Bn, g.OH, g.OW, pd.N, ld=pad8(pd.N), zero=(pad8(pd.N) != pd.N))
        ops.conv_gemm(x.t, wf, g, pd.N, lda=x.ld, a_coff=x.coff, ldb=wf.shape[1], bias=bias, act=act,
                      resid=None if resid is None else resid.t, ldr=None if resid is None else resid.ld,
                      r_coff=0 if resid is None else resid.coff, out=out.t, ldc=out.ld, c_coff=out.coff)
        return out

    # -- layer primitives --------------------------------------------------------------------------------------------
    def gemm(self, x: Act, wname: str, N: int, *, k=1, pad=0, stats=False, geom: Optional[Geom] = None, out: Optional[Act] = None,
             no_dgrad=False, **kw):
        if self.training or not self.fold_bn or not stats:
            return super().gemm(x, wname, N, k=k, pad=pad, stats=stats, geom=geom, out=out, no_dgrad=no_dgrad, **kw)
        for name in ("group", "group_bwd", "variant"):      # launch grouping / tile hints of the training schedule: the folded
            kw.pop(name, None)                              # launch is issued by bn(), alone, on the tile the library picks
        assert not kw, "a convolution in front of a BatchNorm has no bias / residual / dropout / transposed copy: %r" % (kw,)
        g = geom or Geom(x.Bn, x.H, x.W, x.C, k, k, 1, pad)
        ph = out if out is not None else Act(None, g.Bn, g.OH, g.OW, N)
        ph.aux["pending"] = _Pending(x, wname, N, g)
        return ph, None

    def bn(self, y: Act, st, pfx: str, *, relu=True, pool=False, ident: Optional[Act] = None, y2: Optional[Act] = None, st2=None,
           pfx2: Optional[str] = None, mul=None, out: Optional[Act] = None, want_stats=False, single_consumer=False):
        pd = y.aux.pop("pending", None) if not self.training else None      # (single_consumer: a hint for the training backward only)
        if pd is None:
            return super().bn(y, st, pfx, relu=relu, pool=pool, ident=ident, y2=y2, st2=st2, pfx2=pfx2, mul=mul, out=out,
                              want_stats=want_stats)
        if mul is not None or want_stats:
            # the BatchNorm output is rescaled per sample before the next layer sees it: plain convolution + the eval-mode
            # apply kernel (neck.f1_v_proj, model/layers.py:289)
            ya = self._launch(pd, self.WF[pd.wname], None, 0, None, y if y.t is not None else None)
            return super().bn(ya, None, pfx, relu=relu, pool=pool, ident=ident, y2=y2, st2=st2, pfx2=pfx2, mul=mul, out=out,
                              want_stats=want_stats)
        wf, shift = self._folded(pd.wname, pfx)
        resid = ident
        if y2 is not None:
            pd2 = y2.aux.pop("pending")
            wf2, shift2 = self._folded(pd2.wname, pfx2)
            resid = self._launch(pd2, wf2, shift2, 0, None, None)            # bnd(convd(xi))
        act = 0 if not relu else (3 if resid is not None else 1)
        if pool:
            full = self._launch(pd, wf, shift, act, resid, None)
            if out is None:
                out = self.new_act(full.Bn, full.H // 2, full.W // 2, full.C)
            ops.avgpool2_fwd(full.t, full.Bn, full.H, full.W, full.C, out.t, ldx=full.ld, xcoff=full.coff, ldy=out.ld, ycoff=out.coff)
            return out
        if out is None and y.t is not None:
            out = y                                                          # the caller's own buffer (stem: gemm(out=y))
        return self._launch(pd, wf, shift, act, resid, out)


class InferenceRunner:
    """Batched inference on one GPU: `runner(img, word)` -> logits [B, 1, S/4, S/4] (fp32, a buffer that the next call
    overwrites); with `upsample=True` -> sigmoid probabilities at the input size [B, S, S] (engine/engine.py:101-106).
    The first call with a new (batch, size, length) runs eagerly (folds the BatchNorms, uploads tables, warms the allocator),
    the second is captured, later ones replay the graph: one host call per batch."""

    def __init__(self, clip: ClipSpec, head: HeadSpec, state_dict, device, fold_bn: bool = True, use_graph: bool = True,
                 upsample: bool = False, tensors=None):
        """state_dict: a reference-keyed state_dict (copied to `device`); or `tensors=(params, buffers)`: dicts of device
        tensors to run on IN PLACE (the drop-in module hands over its own parameters; call `invalidate()` when they change)"""
        from .trainer import split_state_dict
        self.device = device
        params, buffers = tensors if tensors is not None else split_state_dict(state_dict, device)
        self.engine = InferEngine(clip, head, params, buffers, device, fold_bn=fold_bn)
        self.use_graph = use_graph and torch.device(device).type == "cuda"
        self.upsample = upsample
        self._shapes = {}               # shape key -> dict(calls, img, word, graph, out)
        self.graph_error = None

    def invalidate(self):
        """the tensors this runner works on were changed by someone else: fold, pack and capture again"""
        self.engine.invalidate()
        self._shapes = {}

    def load_state_dict(self, sd):
        e = self.engine
        for k, t in list(e.P.items()) + list(e.Bf.items()):
            t.copy_(sd[k].to(self.device))
        e.invalidate()
        self._shapes = {}

    def _body(self, img, word):
        pred = self.engine.forward(img, word, None, training=False)
        if not self.upsample:
            return pred
        from . import evalpost
        return evalpost.sigmoid_upsample(pred, img.shape[-2], img.shape[-1])

    @torch.no_grad()
    def __call__(self, img, word):
        if not self.use_graph:
            return self._body(img, word)
        key = (tuple(img.shape), tuple(word.shape))
        st = self._shapes.get(key)
        if st is None:
            st = self._shapes[key] = dict(calls=0, img=img.clone(), word=word.clone(), graph=None, out=None)
        st["calls"] += 1
        st["img"].copy_(img, non_blocking=True)
        st["word"].copy_(word, non_blocking=True)
        if st["calls"] == 1:
            return self._body(st["img"], st["word"])
        if st["graph"] is None and self.graph_error is None:
            try:
                from . import capture
                g = torch.cuda.CUDAGraph()
                with capture.graph(g, device=self.device):
                    out = self._body(st["img"], st["word"])
                st["graph"], st["out"] = g, out
            except Exception as ex:          # noqa: BLE001
                self.graph_error = repr(ex)
                torch.cuda.synchronize(self.device)
        if st["graph"] is None:
            return self._body(st["img"], st["word"])
        st["graph"].replay()
        return st["out"]
