"""CRIS forward / backward executor on the HIP library.

`Engine` owns the *schedule* of the CRIS training path - which libcris_hip.so launcher runs on which
buffers, in which order - and a reverse tape for the backward pass.  It does no arithmetic itself: torch
provides device buffers and the current stream, every FLOP is in csrc/.  Layout: activations NHWC bf16
([B*H*W, C] row-major; channel slices of wider buffers replace torch.cat), transformer residual streams
fp32, parameters fp32 (bf16 GEMM-layout copies are re-packed once per step by batched launches, one per arena stage).

Reference call graph being restated (file:line in DerrickWang005/CRIS.pytorch):
  CRIS.forward model/segmenter.py:29-62 -> ModifiedResNet.forward model/clip.py:207-223 (Bottleneck :44-57,
  AttentionPool2d :110-144) ; CLIP.encode_text model/clip.py:439-456 ; FPN.forward model/layers.py:282-309 ;
  TransformerDecoder.forward model/layers.py:154-188 (layer :224-250) ; Projector.forward model/layers.py:63-84.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch
from torch import nn

from . import debug, ops, tables
from .arch import ClipSpec, HeadSpec, build_param_tree
from .ops import BF16, Drop, Geom, NO_DROP, pad8, pad32

F32 = torch.float32
BN_EPS, BN_MOM, LN_EPS = 1e-5, 0.1, 1e-5


class Act:
    """A [M, ld] activation buffer (or a channel slice of one) plus its gradient buffer."""
    __slots__ = ("t", "Bn", "H", "W", "C", "ld", "coff", "root", "_g", "aux")
    _engine = None        # the Engine whose step is being scheduled (zero-filled gradient buffers come from its slab)

    def __init__(self, t, Bn, H, W, C, ld=None, coff=0, root=None):
        self.t, self.Bn, self.H, self.W, self.C = t, Bn, H, W, C
        self.ld = ld if ld is not None else C
        self.coff = coff
        self.root = root if root is not None else self
        self._g = None
        self.aux = {}

    @property
    def M(self):
        return self.Bn * self.H * self.W

    @property
    def g(self):
        return self.root._g

    def slice(self, coff, C):
        return Act(self.t, self.Bn, self.H, self.W, C, self.ld, self.coff + coff, self.root)

    def grad_target(self):
        """(grad tensor, already_holds_a_gradient)."""
        r = self.root
        if r._g is None:
            # a channel slice writes only its columns: the rest must read as zero for later accumulation
            partial = self.C < r.t.shape[-1]
            if partial:
                eng = Act._engine
                r._g = eng.zeros(*r.t.shape, dtype=r.t.dtype) if eng is not None else ops.zero_(torch.empty_like(r.t))
            else:
                r._g = torch.empty_like(r.t)
            return r._g, partial
        return r._g, True


class GemmGroup:
    """forward (f) and backward (b) launch queues of one group of independent GEMM layers (Engine.group_begin)"""
    __slots__ = ("f", "b")

    def __init__(self):
        self.f, self.b = ops.GemmQueue(), ops.GemmQueue()


class Comm:
    """Cross-rank hooks used by SyncBN / gradient averaging (dist.py provides the RCCL implementation)."""
    world = 1

    def allreduce_sum(self, t):  # pragma: no cover - single GPU default
        return

    def begin_step(self):
        """Called once per scheduled training step before the first exchange (exchange numbering restarts)."""

    def broadcast(self, t, src=0):  # pragma: no cover - single GPU default
        return

    def all_gather_object(self, obj):
        return [obj]

    def allreduce_sum_op(self, t):
        """The exchange of `t` as a callable bound NOW (schedule time) - what ops.torch_op runs and a command list replays."""
        return lambda: self.allreduce_sum(t)

    def next_link(self):
        """hip.P2PLink of the next SyncBN exchange of the step when the peer mailboxes are in use (the BatchNorm kernels then
        exchange inside their own launches), else None (exchange by allreduce_sum_op)"""
        return None


class Engine:
    def __init__(self, clip: ClipSpec, head: HeadSpec, params: Dict[str, torch.Tensor], buffers: Dict[str, torch.Tensor],
                 device, comm: Optional[Comm] = None, sync_bn: bool = False, inference_only: bool = False):
        self.clip, self.head, self.dev = clip, head, device
        # inference_only (infer.InferEngine): no gradient arena (587 MB fp32 at R50) and no input-gradient weight packs -
        # a model that evaluates during training (engine/engine.py:90-123 `validate`) would otherwise hold both twice
        self.inference_only = inference_only
        self.embed_live = None          # optional uint8 [vocabulary]: rows of the token embedding that ever had a gradient (trainer)
        # CRIS_STATE_FP32=1: the sentence vector and the (at most batch-size rows of) layers it runs through in fp32
        # (csrc/smallf32.hip).  Measured (calls r04d / r04f, profiles/parity_r04.md): logits 15 % closer to the fp32 oracle and better
        # gradient cosines over the 100 teacher-forced states, no change of their mean loss error, +0.19 ms per step - off by default
        import os
        self.state_f32 = os.environ.get("CRIS_STATE_FP32", "0") == "1"
        self.P, self.Bf = params, buffers
        self.comm = comm or Comm()
        self.sync_bn = sync_bn and (self.comm.world > 1 or debug.HOOKS.force_dist)
        self.tape: List[Callable[[], None]] = []
        self.training = True
        self.seed = 0
        self.seed_dev = None            # optional device word added to every dropout seed (graph replay: trainer.py)
        # the text encoder (12 layers of M = B*L ~ 136-row GEMMs: latency-bound, ~16 workgroups each) is independent of
        # the visual encoder until the neck: it runs on a second HIP stream, forward and backward, underneath the convs
        self.side = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None
        if debug.HOOKS.no_side_stream:
            self.side = None
        # weight gradients of the mid-size layers are queued and launched together at arena-stage boundaries (ops.WgradQueue)
        self._wq, self._sq = ops.WgradQueue(self._flush_wgrads), ops.SumQueue()
        self._zslab, self._zcur, self._zneed, self._zneed_last = None, 0, 0, 0
        self._tables = {}
        self._build_grad_arena()
        self._build_packs()

    # ------------------------------------------------------------------------------------------
    # parameter-side setup
    # ------------------------------------------------------------------------------------------
    def gemm_layout(self, name):
        """(N, Cin, taps, Cpad) of a weight that runs through the GEMM kernels, or None.  Its gradient is kept in the
        GEMM layout [N][taps*Cpad] (k = tap*Cpad + c), which for taps == 1 and Cpad == Cin IS the parameter layout."""
        p = self.P[name]
        if name == "backbone.visual.conv1.weight":
            return p.shape[0], 27, 1, 32                     # stem: im2col'd, k = ci*9 + kh*3 + kw (parameter order)
        if p.dim() == 4:
            return p.shape[0], p.shape[1], p.shape[2] * p.shape[3], pad8(p.shape[1])
        return None

    def grad_param_layout(self, name):
        """Gradient of `name` as a tensor shaped like the parameter (a relayout copy for 3x3 / padded convolutions; used by
        tests, tools and checkpoint-style export - the optimizer reads the GEMM layout directly)."""
        g, lay = self.G[name], self.gemm_layout(name)
        if lay is None or (lay[2] == 1 and lay[3] == lay[1]):
            return g.view(self.P[name].shape)
        N, Cin, taps, Cpad = lay
        return g.view(N, taps, Cpad)[:, :, :Cin].permute(0, 2, 1).reshape(self.P[name].shape)

    def grad_param_view(self, name):
        """the same as a VIEW of the arena, never a copy (raises where the layout does not allow one): what the drop-in module
        hands out as `.grad` when cris.pytorch_amd.optim.Adam reads the arena directly"""
        g, lay = self.G[name], self.gemm_layout(name)
        if lay is None or (lay[2] == 1 and lay[3] == lay[1]):
            return g.view(self.P[name].shape)
        N, Cin, taps, Cpad = lay
        return g.view(N, taps, Cpad)[:, :, :Cin].permute(0, 2, 1).view(self.P[name].shape)

    def grads_param_layout(self):
        return {k: self.grad_param_layout(k) for k in self.G}

    def _build_grad_arena(self):
        """One flat fp32 gradient arena; BatchNorm blocks are laid out [dbeta | dgamma] (and
        [bn3 | downsample.1] pairs contiguous) so bn_bwd_reduce accumulates straight into them."""
        tree = build_param_tree(self.clip, self.head)
        bn_prefixes = [n for n, m in tree.named_modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
        self.bn_prefixes = bn_prefixes
        pairs = {}
        for n, m in tree.named_modules():
            if m.__class__.__name__ == "BottleneckP" and m.downsample is not None:
                pairs[n + ".bn3"] = n + ".downsample.1"
        paired_second = set(pairs.values())
        bn_set = set(bn_prefixes)
        # forward-compute order (visual, text, neck, decoder, proj) so that backward completes the arena from its
        # end towards its start and contiguous suffixes can be all-reduced while earlier layers still run
        def stage(name):
            # arena stages = units of the gradient exchange: 0 stem+layer1, 1 layer2, 2 layer3, 3 layer4+attnpool, 4 text,
            # 5 neck, 6 decoder, 7 projector.  Backward finishes them 7, 6, 5, then 4 (side stream) and 3, 2, 1, 0.
            if name.startswith("backbone.visual."):
                rest = name[len("backbone.visual."):]
                if rest.startswith("layer2"):
                    return 1
                if rest.startswith("layer3"):
                    return 2
                if rest.startswith(("layer4", "attnpool")):
                    return 3
                return 0
            if name.startswith("backbone."):
                return 4
            return {"neck": 5, "decoder": 6, "proj": 7}[name.split(".")[0]]
        # Inside a stage: first the gradients that are ACCUMULATED into or only partly written each step (BatchNorm sums,
        # embedding tables: rows of absent tokens / positions must read 0) - they are zeroed at the start of a step, one
        # contiguous range per stage - then the ones a kernel overwrites completely every step (GEMM weights and their
        # biases: cris_conv_wgrad stores; LayerNorm: cris_sum_tables stores), which are never zeroed (~480 of 587 MB).
        def needs_zero(name):
            pfx = name.rsplit(".", 1)[0]
            return pfx in bn_set or "embedding" in name or name == "backbone.logit_scale"
        self.needs_zero = needs_zero
        names = sorted(self.P.keys(), key=lambda k: (stage(k), 0 if needs_zero(k) else 1))   # stable: module order otherwise
        order, seen = [], set()
        for name in names:
            if name in seen:
                continue
            pfx = name.rsplit(".", 1)[0]
            if pfx in bn_set:
                if pfx in paired_second:
                    continue                                           # emitted together with its bn3 partner
                for g in [pfx] + ([pairs[pfx]] if pfx in pairs else []):
                    order += [g + ".bias", g + ".weight"]
                    seen.update((g + ".bias", g + ".weight"))
            else:
                order.append(name)
                seen.add(name)
        assert set(order) == set(self.P.keys())
        self.stage_of = stage
        total, offs, shapes = 0, {}, {}
        for name in order:
            lay = self.gemm_layout(name)
            shapes[name] = tuple(self.P[name].shape) if lay is None else (lay[0], lay[2] * lay[3])
            n = 1
            for d in shapes[name]:
                n *= d
            offs[name] = (total, n)
            total += (n + 3) // 4 * 4            # keep every view 16-byte aligned
        if self.inference_only:
            self.grad_arena, self.G = None, {name: None for name in offs}
            self.grad_order, self.grad_offsets, self.bn_pairs, self.stage_ranges, self.zero_ranges = order, offs, pairs, {}, []
            return
        self.grad_arena = torch.zeros(total, dtype=F32, device=self.dev)
        self.G = {name: self.grad_arena[o:o + n].view(shapes[name]) for name, (o, n) in offs.items()}
        self.grad_order = order
        self.grad_offsets = offs
        self.bn_pairs = pairs
        # arena [start, end) of each stage, and of the part of it that is zeroed every step
        self.stage_ranges, zr = {}, {}
        for name in order:
            st = stage(name)
            lo, hi = offs[name][0], offs[name][0] + (offs[name][1] + 3) // 4 * 4
            a, b = self.stage_ranges.get(st, (lo, hi))
            self.stage_ranges[st] = (min(a, lo), max(b, hi))
            if needs_zero(name):
                a, b = zr.get(st, (lo, hi))
                zr[st] = (min(a, lo), max(b, hi))
        self.zero_ranges = [self.grad_arena[a:b] for a, b in sorted(zr.values())]

    def _add_pack(self, name, N, Cin, taps, Cpad=None, want_D=True, transposed=False):
        src = self.P[name]
        tab = self.packs[self.stage_of(name)]
        f, d = tab.add(src, N, Cin, taps, Cpad=Cpad, want_D=want_D, src_transposed=transposed)
        self.WF[name], self.WD[name] = f, d
        self.pack_info[name] = tab.info[-1]

    def _build_packs(self):
        # bf16 GEMM-operand copies of the weights (F: forward, D: input gradient).  The native trainer's Adam rewrites them
        # from the updated values (packs_current stays True); a forward that finds them stale - first step, or parameters
        # changed by someone else (the drop-in module under a torch optimizer) - re-packs all with one launch per stage
        self.packs = {st: ops.PackTable() for st in range(8)}
        self.packs_current = False
        self.WF, self.WD, self.pack_info = {}, {}, {}
        wd = not self.inference_only
        for name, p in self.P.items():
            if p.dim() == 4:
                N, Cin, taps, Cpad = self.gemm_layout(name)
                self._add_pack(name, N, Cin, taps, Cpad=Cpad, want_D=wd and name != "backbone.visual.conv1.weight")
            elif name == "backbone.text_projection":
                self._add_pack(name, p.shape[1], p.shape[0], 1, want_D=wd, transposed=True)      # used as x @ P
            elif p.dim() == 2 and name.endswith(("weight", "in_proj_weight")) and "embedding" not in name:
                self._add_pack(name, p.shape[0], p.shape[1], 1, want_D=wd)
        for t in self.packs.values():
            if t.descs:
                t.finalize(self.dev)

    def repack_weights(self):
        for t in self.packs.values():
            t.run()

    def repack_stage(self, st):
        self.packs[st].run()

    def table(self, key, fn):
        if key not in self._tables:
            self._tables[key] = torch.from_numpy(fn()).to(self.dev)
        return self._tables[key]

    # ------------------------------------------------------------------------------------------
    # small helpers
    # ------------------------------------------------------------------------------------------
    # Zero-filled temporaries (padded outputs, partial-slice gradient buffers, small accumulators: ~70 per step) are carved
    # from one slab that is cleared by a single launch at the start of the step; its size is what the previous step used.
    def _zero_slab_begin(self):
        need = self._zneed_last
        if need and (self._zslab is None or self._zslab.numel() < need):
            self._zslab = torch.empty(need, dtype=torch.uint8, device=self.dev)
        if self._zslab is not None:
            ops.zero_(self._zslab)
        self._zcur = self._zneed = 0

    def zeros(self, *shape, dtype=F32):
        n = dtype.itemsize
        for d in shape:
            n *= int(d)
        n_al = (n + 255) // 256 * 256
        self._zneed += n_al
        if n > 0 and self._zslab is not None and self._zcur + n_al <= self._zslab.numel():
            t = self._zslab[self._zcur:self._zcur + n].view(dtype).view(*shape)
            self._zcur += n_al
            return t
        return ops.zero_(torch.empty(*shape, dtype=dtype, device=self.dev))

    def empty(self, *shape, dtype=BF16):
        return torch.empty(*shape, dtype=dtype, device=self.dev)

    def new_act(self, Bn, H, W, C, ld=None, dtype=BF16, zero=False):
        ld = ld if ld is not None else C
        t = (self.zeros if zero else self.empty)(Bn * H * W, ld, dtype=dtype)
        return Act(t, Bn, H, W, C, ld)

    def _flush_wgrads(self):
        self._wq.flush()

    def _flush_queues(self):
        """launch what backward has queued so far on the current stream: grouped weight gradients and the ordered sums of the
        LayerNorm parameter-gradient partials"""
        self._wq.flush()
        self._sq.flush()

    def drop(self, layer, site):
        p = self.head.dropout if self.training else 0.0
        return Drop(p, self.seed, layer * 8 + site, self.seed_dev) if p > 0 else NO_DROP

    # ------------------------------------------------------------------------------------------
    # GEMM layer (conv / linear) forward + tape
    # ------------------------------------------------------------------------------------------
    def gemm(self, x: Act, wname: str, N: int, *, k=1, pad=0, rows=None, bias: Optional[str] = None, out: Optional[Act] = None,
             out_f32=False, resid: Optional[Act] = None, drop: Drop = NO_DROP, stats=False, outT=None, geom: Optional[Geom] = None,
             no_dgrad=False, stream_grad: Optional[Act] = None, w_transposed=False, group: Optional["GemmGroup"] = None,
             group_bwd=True, variant=-1):
        """y = conv_k(x) with weight `wname` (rows n0:n1 of it when `rows`), optional bias / residual / dropout /
        BN statistics / transposed head-split copy.  `stream_grad`: fp32 residual-stream Act whose gradient is the
        gradient of this layer's output (post dropout) - used for `x + dropout(linear(..))` branches.
        `group` (see group_begin): the forward launch waits for group_end() and runs together with the group's other members;
        with `group_bwd` its input-gradient GEMM does the same in backward (only when no other member writes x's gradient).
        `variant`: tile variant forced for the forward and input-gradient launches (grouped launches share one tile)."""
        g = geom or Geom(x.Bn, x.H, x.W, x.C, k, k, 1, pad)
        Wf, Wd = self.WF[wname], self.WD.get(wname)
        Gw = self.G[wname]
        n0 = 0
        if rows is not None:
            n0, n1 = rows
            Wf = Wf[n0:n1]
            Gw = Gw[n0:n1] if Gw is not None else None
        ldbF = self.WF[wname].shape[1]
        bias_t = None
        if bias is not None:
            bias_t = self.P[bias][n0:n0 + N]
        if out is None:
            out = self.new_act(g.Bn, g.OH, g.OW, N, ld=pad8(N), dtype=F32 if out_f32 else BF16, zero=(pad8(N) != N))
        kw = {}
        if outT is not None:
            kw = dict(outT=outT["buf"], T_L=outT["L"], T_Lpad=outT["Lpad"], T_E=outT["E"], T_sec_stride=outT["sec_stride"])
        st = ops.conv_gemm(x.t, Wf, g, N, lda=x.ld, a_coff=x.coff, ldb=ldbF, bias=bias_t,
                           resid=None if resid is None else resid.t, ldr=None if resid is None else resid.ld,
                           r_coff=0 if resid is None else resid.coff, out=out.t, ldc=out.ld, c_coff=out.coff,
                           stats=stats, drop=drop, variant=variant, queue=None if group is None else group.f, **kw)
        if not self.training:
            return (out, st) if stats else out

        def bwd():
            if stream_grad is not None:
                gy = self.empty(out.M, pad8(N))
                ops.cast_f32_bf16_drop(stream_grad.g, gy, drop)
                gy_ld, gy_coff = pad8(N), 0
            else:
                gy, gy_ld, gy_coff = out.g, out.ld, out.coff
            if w_transposed:
                # parameter stored [in, out] (used as x @ P): dP = x^T dY - same kernel with the operand roles swapped
                ops.conv_wgrad(x.t, gy, Geom.linear(out.M, pad8(N)), x.C, Gw, ldy=x.ld, y_coff=x.coff, N_ld=x.C, ldx=gy_ld,
                               x_coff=gy_coff, queue=self._wq)
            else:
                ops.conv_wgrad(gy, x.t, g, N, Gw, ldy=gy_ld, y_coff=gy_coff, N_ld=pad8(N), ldx=x.ld, x_coff=x.coff,
                               dbias=None if bias is None else self.G[bias][n0:n0 + N], queue=self._wq)
            if no_dgrad:
                return
            assert N % 8 == 0, "dgrad path needs N % 8 == 0 (pad the gradient buffer otherwise)"
            gD = Geom(g.Bn, g.OH, g.OW, N, k, k, 1, pad)            # stride-1 'same' conv: dgrad = conv of dY with flipped taps
            wd = Wd[:, n0:] if rows is not None else Wd
            gx, acc = x.grad_target()
            tkw = {}
            dT = self._dgrad_outT
            if dT is not None:
                tkw = dict(outT=dT["buf"], T_L=dT["L"], T_Lpad=dT["Lpad"], T_E=dT["E"], T_sec_stride=dT["sec_stride"])
            q = group.b if (group is not None and group_bwd) else None
            src = x.aux.get("bn_src")
            if src is not None and ops.BNR_FUSE and not acc and q is None and dT is None and x.root is x:
                # x = relu(bn(y)) and this layer is its ONLY consumer (Engine.bn(single_consumer=True)): the gradient written
                # here is complete, so the epilogue also produces the BatchNorm backward's partial sums (sum g, sum g xhat)
                src["parts"] = ops.conv_gemm(gy, wd, gD, Wd.shape[0], lda=gy_ld, a_coff=gy_coff, ldb=Wd.shape[1], out=gx, ldc=x.ld,
                                             c_coff=x.coff, variant=variant, bnr=src)
            else:
                ops.conv_gemm(gy, wd, gD, Wd.shape[0], lda=gy_ld, a_coff=gy_coff, ldb=Wd.shape[1], out=gx, ldc=x.ld, c_coff=x.coff,
                              resid=gx if acc else None, ldr=x.ld, r_coff=x.coff, variant=variant, queue=q, **tkw)

        self.tape.append(bwd)
        return (out, st) if stats else out

    def group_begin(self) -> "GemmGroup":
        """Open a group of INDEPENDENT GEMM layers: gemm(..., group=G) defers its forward launch to group_end(G), where the
        members run as grouped launches (ops.GemmQueue: one launch per tile variant / epilogue kind present); in backward the
        members' input-gradient GEMMs wait for the closure appended here, which runs after theirs."""
        G = GemmGroup()
        if self.training:
            self.tape.append(G.b.flush)
        return G

    def group_end(self, G: "GemmGroup"):
        G.f.flush()

    @staticmethod
    def _group_variant(site, default):
        """tile variant forced on the members of a group whose own choices differ (CRIS_GROUP_VARIANT_<SITE> overrides; "auto":
        every member keeps its own tile and only equal ones share a launch)"""
        import os
        v = os.environ.get("CRIS_GROUP_VARIANT_" + site.upper(), default)
        return -1 if v == "auto" else v

    # ------------------------------------------------------------------------------------------
    # BatchNorm layer
    # ------------------------------------------------------------------------------------------
    def _bn_coeffs(self, pfx, st, count, C):
        """scale/shift (+ saved mean/invstd) of one BatchNorm from the statistics partials `st` [2][nparts][C]."""
        scale, shift, mean, invstd = self.empty(C, dtype=F32), self.empty(C, dtype=F32), self.empty(C, dtype=F32), self.empty(C, dtype=F32)
        gamma, beta = self.P[pfx + ".weight"], self.P[pfx + ".bias"]
        rm, rv = self.Bf[pfx + ".running_mean"], self.Bf[pfx + ".running_var"]
        if not self.training:
            ops.bn_eval_coeffs(gamma, beta, rm, rv, BN_EPS, C, scale, shift)
            return scale, shift, mean, invstd, count
        if self.sync_bn:
            # SyncBatchNorm (train.py:97-98): ONE exchange of [S1 | S2] (moments about the running mean, which every rank
            # holds identically) instead of torch's all_gather of (mean, invstd, count)
            gcount = count * self.comm.world
            link = self.comm.next_link()
            if link is not None:
                # peer mailboxes: the exchange happens INSIDE the finalize launch (one launch, as without SyncBN)
                ops.bn_finalize(st, count, gcount, gamma, beta, rm, rv, BN_MOM, BN_EPS, C, scale, shift, mean, invstd, link=link)
                return scale, shift, mean, invstd, gcount
            merged = self.zeros(2 * C)
            ops.bn_finalize(st, count, count, gamma, beta, None, None, BN_MOM, BN_EPS, C, None, None, mean, None, merged=merged)
            ops.bn_sync_pack(merged, mean, rm, count, C)
            ops.torch_op(self.comm.allreduce_sum_op(merged))
            ops.bn_sync_unpack(merged, rm, gcount, C)
            ops.bn_finalize(None, count, gcount, gamma, beta, rm, rv, BN_MOM, BN_EPS, C, scale, shift, mean, invstd,
                            global_stats=merged)
            return scale, shift, mean, invstd, gcount
        ops.bn_finalize(st, count, count, gamma, beta, rm, rv, BN_MOM, BN_EPS, C, scale, shift, mean, invstd)
        return scale, shift, mean, invstd, count

    def bn(self, y: Act, st, pfx: str, *, relu=True, pool=False, ident: Optional[Act] = None, y2: Optional[Act] = None, st2=None,
           pfx2: Optional[str] = None, mul=None, out: Optional[Act] = None, want_stats=False, single_consumer=False):
        """z = [pool](relu(bn(y) [+ bn2(y2)] [+ ident]) [* mul]); training statistics come from the conv epilogue.
        single_consumer: the caller promises that z feeds exactly ONE gemm layer - that layer's input-gradient GEMM may then
        produce this BatchNorm's backward partial sums in its epilogue (model/clip.py:47-50: bn1 -> conv2, bn2 -> conv3)."""
        C = y.C
        count = float(y.M)
        scale, shift, mean, invstd, gcount = self._bn_coeffs(pfx, st, count, C)
        sc2 = sh2 = mean2 = inv2 = None
        if y2 is not None:
            sc2, sh2, mean2, inv2, _ = self._bn_coeffs(pfx2, st2, count, C)
        OH, OW = (y.H // 2, y.W // 2) if pool else (y.H, y.W)
        if out is None:
            out = self.new_act(y.Bn, OH, OW, C)
        ops.bn_apply(y.t, scale, shift, out.t, y.Bn, y.H, y.W, C, ldy=y.ld, y_coff=y.coff, ldz=out.ld, z_coff=out.coff, relu=relu,
                     pool=pool, y2=None if y2 is None else y2.t, ldy2=None if y2 is None else y2.ld,
                     y2_coff=0 if y2 is None else y2.coff, scale2=sc2, shift2=sh2, ident=None if ident is None else ident.t,
                     ldi=None if ident is None else ident.ld, i_coff=0 if ident is None else ident.coff, mul=mul)
        if want_stats:
            # statistics of this output for a BatchNorm that follows without a conv in between (FPN norm_layer)
            out.aux["stats"] = ops.colstats(out.t, out.M, C, 32, self.dev, ldx=out.ld, coff=out.coff)
        if not self.training:
            return out
        dmul = self.empty(y.Bn, C, dtype=F32) if mul is not None else None
        out.aux["dmul"] = dmul
        src = None
        if single_consumer and relu and not pool and ident is None and y2 is None and mul is None and out.root is out:
            src = dict(y=y.t, ldy=y.ld, coff=y.coff, mean=mean, invstd=invstd, scale=scale, shift=shift, parts=None)
            out.aux["bn_src"] = src

        def bwd():
            Gb = self.G[pfx + ".bias"]
            # [dbeta | dgamma] (and the paired downsample block) are contiguous in the arena
            nblk = 4 * C if y2 is not None else 2 * C
            arena_block = self.grad_arena[Gb.storage_offset():Gb.storage_offset() + nblk]
            link = self.comm.next_link() if self.sync_bn else None
            if link is not None:
                sums = self.empty(nblk, dtype=F32)          # sums over all ranks, written by the summation launch itself
            elif self.sync_bn:
                sums = self.zeros(nblk)
            else:
                sums = arena_block
            dy, _ = y.grad_target()
            dy2 = None
            if y2 is not None:
                dy2, _ = y2.grad_target()
            did = did_acc = None
            if ident is not None:
                did, did_acc = ident.grad_target()

            def between(s):
                ops.axpy_f32(arena_block, s, 1.0)
                ops.torch_op(self.comm.allreduce_sum_op(s))

            need_z = relu and not pool and (ident is not None or y2 is not None)
            ops.bn_bwd(out.g, y.t, scale, shift, mean, invstd, sums, dy, y.Bn, y.H, y.W, C, gcount, lddz=out.ld, dz_coff=out.coff,
                       ldy=y.ld, y_coff=y.coff, lddy=y.ld, dy_coff=y.coff, relu=relu, pool=pool,
                       z=out.t if need_z else None, ldz=out.ld, z_coff=out.coff,
                       y2=None if y2 is None else y2.t, ldy2=None if y2 is None else y2.ld, y2_coff=0 if y2 is None else y2.coff,
                       mean2=mean2, invstd2=inv2, scale2=sc2, dy2=dy2, lddy2=None if y2 is None else y2.ld,
                       dy2_coff=0 if y2 is None else y2.coff, mul=mul, dmul=dmul, dident=did,
                       lddi=None if ident is None else ident.ld, di_coff=0 if ident is None else ident.coff,
                       dident_accum=bool(did_acc), between=between if (self.sync_bn and link is None) else None,
                       link=link, local_sums=arena_block if link is not None else None,
                       pre_reduced=None if src is None else src["parts"])

        self.tape.append(bwd)
        return out

    def conv_bn(self, x: Act, pfx_conv: str, pfx_bn: str, N: int, k=1, pad=0, relu=True, pool=False, out=None, single_consumer=False, **kw):
        y, st = self.gemm(x, pfx_conv + ".weight", N, k=k, pad=pad, stats=True, **kw)
        return self.bn(y, st, pfx_bn, relu=relu, pool=pool, out=out, single_consumer=single_consumer)

    # ------------------------------------------------------------------------------------------
    # LayerNorm layer
    # ------------------------------------------------------------------------------------------
    def ln(self, x: Act, pfx: str, *, want_y=True, pos=None, pos_rows=0, resid: Optional[Act] = None, out_drop: Drop = NO_DROP,
           in_relu=False, in_drop: Drop = NO_DROP, dx_stream: Optional[Act] = None):
        """Returns (y, ypos, out_stream).  `dx_stream`: fp32 residual-stream Act that receives dx (+=) in backward
        (pre-norm sites); otherwise dx goes to x's own bf16 gradient buffer."""
        rows, C = x.M, x.C
        gamma, beta = self.P[pfx + ".weight"], self.P[pfx + ".bias"]
        mean, rstd = self.empty(rows, dtype=F32), self.empty(rows, dtype=F32)
        y = self.new_act(x.Bn, x.H, x.W, C) if want_y else None
        ypos = self.new_act(x.Bn, x.H, x.W, C) if pos is not None else None
        outs = self.new_act(x.Bn, x.H, x.W, C, dtype=F32) if resid is not None else None
        ops.ln_fwd(x.t, gamma, beta, rows, C, mean, rstd, ldx=x.ld, y=None if y is None else y.t,
                   ypos=None if ypos is None else ypos.t, pos=pos, pos_rows=pos_rows, resid=None if resid is None else resid.t,
                   out_f32=None if outs is None else outs.t, in_relu=in_relu, in_drop=in_drop, out_drop=out_drop, eps=LN_EPS)
        if y is not None:
            y.aux["ln_stats"] = (mean, rstd)
        if self.training:
            def bwd():
                if outs is not None:
                    # identity path of the residual stream: d(resid) = d(out)
                    if resid.root._g is None:
                        resid.root._g = outs.g
                    elif resid.root._g is not outs.g:
                        ops.axpy_f32(resid.root._g, outs.g, 1.0)
                if dx_stream is not None:
                    dx, accum = dx_stream.grad_target()
                else:
                    dx, acc = x.grad_target()
                    accum = False
                    assert not acc, "bf16 LN input gradient cannot be accumulated"
                ops.ln_bwd(x.t, gamma, mean, rstd, rows, C, dx, ldx=x.ld, dy=None if (y is None or y.g is None) else y.g,
                           dypos=None if (ypos is None or ypos.g is None) else ypos.g,
                           dout_f32=None if outs is None else outs.g, dgamma=self.G[pfx + ".weight"], dbeta=self.G[pfx + ".bias"],
                           dx_accum=accum, in_relu=in_relu, in_drop=in_drop, out_drop=out_drop, queue=self._sq)

            self.tape.append(bwd)
        return y, ypos, outs

    # ------------------------------------------------------------------------------------------
    # attention layer
    # ------------------------------------------------------------------------------------------
    def new_T(self, B, Hn, L, secs=1):
        Lpad = pad32(L)
        buf = self.zeros(secs, B * Hn * 64, Lpad, dtype=BF16)
        return dict(buf=buf, L=L, Lpad=Lpad, E=Hn * 64, sec_stride=B * Hn * 64 * Lpad)

    def attention(self, q: Act, k: Act, v: Act, Qt, Kt, Vt, B, Hn, Lq, Lk, *, causal=False, key_tokens=None, drop: Drop = NO_DROP):
        """q/k/v: Acts whose (t, ld, coff) address [B*L, Hn*64] operands; Qt/Kt/Vt: [B*Hn*64, Lpad] transposed copies."""
        E = Hn * 64
        o = self.new_act(B, Lq, 1, E)
        lse = self.empty(B * Hn, Lq, dtype=F32)
        Lkp, Lqp = pad32(Lk), pad32(Lq)
        qv, kv, vv = q.t[:, q.coff:], k.t[:, k.coff:], v.t[:, v.coff:]
        prm = ops.attn_params(qv, kv, vv, Vt, B, Hn, Lq, Lk, Lkp, 64 ** -0.5, ldq=q.ld, ldk=k.ld, ldv=v.ld, Kt=Kt, Qt=Qt,
                              Lq_pad=Lqp, key_tokens=key_tokens, causal=causal, drop=drop)
        ops.attn_fwd(prm, o.t, lse, ldo=o.ld)
        if self.training:
            dOt = self.new_T(B, Hn, Lq)
            keep = (qv, kv, vv, Qt, Kt, Vt, key_tokens)

            def bwd():
                dq, aq = q.grad_target()
                dk, ak = k.grad_target()
                dv, av = v.grad_target()
                delta = self.empty(B * Hn, Lq, dtype=F32)
                ops.attn_bwd(prm, o.t, lse, o.g, dOt["buf"], delta, dq[:, q.coff:], dk[:, k.coff:], dv[:, v.coff:], ldo=o.ld,
                             lddo=o.ld, lddq=q.ld, lddk=k.ld, lddv=v.ld)
                _ = keep

            self.tape.append(bwd)
            return o, dOt
        return o, None

    # ------------------------------------------------------------------------------------------
    # network pieces
    # ------------------------------------------------------------------------------------------
    def _bottleneck(self, x: Act, p: str, planes: int, stride: int, has_ds: bool) -> Act:
        if has_ds:
            # conv1 and the downsample convolution (model/clip.py:44-53) read the block's input and nothing of each other: one
            # grouped launch forward; backward too when the downsample branch goes through the average pool (stride 2) - with
            # stride 1 both input gradients accumulate into the SAME buffer, so they stay two launches there
            xi = x
            if stride > 1:
                xi = self.new_act(x.Bn, x.H // 2, x.W // 2, x.C)
                ops.avgpool2_fwd(x.t, x.Bn, x.H, x.W, x.C, xi.t, ldx=x.ld, xcoff=x.coff)
                if self.training:
                    def bwd(xi=xi):
                        gx, acc = x.grad_target()
                        ops.avgpool2_bwd(xi.g, x.Bn, x.H, x.W, x.C, gx, lddx=x.ld, dxcoff=x.coff, accum=acc)
                    self.tape.append(bwd)            # (backward: after the group's flush below, i.e. after conv1's dgrad wrote x.g)
            G = self.group_begin()
            y1, st1 = self.gemm(x, p + ".conv1.weight", planes, stats=True, group=G, group_bwd=stride > 1)
            yd, std = self.gemm(xi, p + ".downsample.0.weight", planes * 4, stats=True, group=G, group_bwd=stride > 1)
            self.group_end(G)
            a1 = self.bn(y1, st1, p + ".bn1", single_consumer=True)
        else:
            a1 = self.conv_bn(x, p + ".conv1", p + ".bn1", planes, single_consumer=True)
        a2 = self.conv_bn(a1, p + ".conv2", p + ".bn2", planes, k=3, pad=1, pool=stride > 1, single_consumer=True)
        y3, st3 = self.gemm(a2, p + ".conv3.weight", planes * 4, stats=True)
        if has_ds:
            return self.bn(y3, st3, p + ".bn3", relu=True, y2=yd, st2=std, pfx2=p + ".downsample.1")
        return self.bn(y3, st3, p + ".bn3", relu=True, ident=x)

    def _encode_image(self, img):
        v = "backbone.visual"
        B, _, H, W = img.shape
        w = self.clip.vision_width
        col = self.empty(B * (H // 2) * (W // 2), 32)
        ops.stem_im2col(img, col)
        xcol = Act(col, B, H // 2, W // 2, 32)
        y = self.new_act(B, H // 2, W // 2, w // 2)
        _, st = self.gemm(xcol, v + ".conv1.weight", w // 2, geom=Geom.linear(xcol.M, 32), stats=True, no_dgrad=True, out=y)
        x = self.bn(y, st, v + ".bn1")
        x = self.conv_bn(x, v + ".conv2", v + ".bn2", w // 2, k=3, pad=1)
        x = self.conv_bn(x, v + ".conv3", v + ".bn3", w, k=3, pad=1, pool=True)
        feats = []
        inpl = w
        for li, nblk in enumerate(self.clip.vision_layers):
            if li > 0:
                self._vis_stage_start[li] = len(self.tape)      # arena stage li = this layer group (3 also takes attnpool)
            planes = w * (1, 2, 4, 8)[li]
            for bi in range(nblk):
                stride = 2 if (li > 0 and bi == 0) else 1
                has_ds = bi == 0 and (stride > 1 or inpl != planes * 4)
                x = self._bottleneck(x, "%s.layer%d.%d" % (v, li + 1, bi), planes, stride, has_ds)
                inpl = planes * 4
            feats.append(x)
        x4 = self._attnpool(feats[3], v + ".attnpool")
        return feats[1], feats[2], x4, feats

    def _attnpool(self, x: Act, p: str) -> Act:
        B, H, W, C = x.Bn, x.H, x.W, x.C
        T, G, Hn = H * W, self.clip.pos_grid, self.clip.vis_heads
        Cout = self.clip.embed_dim
        grp = self.group_begin()
        yc, stc = self.gemm(x, p + ".connect.0.weight", Cout, stats=True, group=grp, group_bwd=False)
        R = self.table(("bicubic", G, H, W), lambda: tables.bicubic_resize_matrix(G, H, W))
        posr = self.empty(T, C, dtype=F32)
        ops.posresize_fwd(R, self.P[p + ".positional_embedding"], T, G, C, posr)
        tok = self.new_act(B, T, 1, C)
        ops.add_rowtable(x.t, posr, T, tok.t, x.M, C, lda=x.ld)
        if self.training:
            def bwd_tok():
                gx, acc = x.grad_target()
                ops.add_bf16(tok.g, gx, x.M, C, b=gx if acc else None, lda=C, ldb=x.ld, bcoff=x.coff, ldy=x.ld, ycoff=x.coff)
                dposr = self.empty(T, C, dtype=F32)
                ops.batch_rowsum(tok.g, B, T, C, dposr)
                ops.posresize_bwd(R, dposr, T, G, C, self.G[p + ".positional_embedding"])
            self.tape.append(bwd_tok)
        Tq, Tk, Tv = self.new_T(B, Hn, T), self.new_T(B, Hn, T), self.new_T(B, Hn, T)
        # q / k / v (+ the `connect` convolution queued above) are independent: grouped launches forward (model/clip.py:112-139);
        # their three input gradients accumulate into ONE buffer (tok.g), so backward keeps them apart
        q = self.gemm(tok, p + ".q_proj.weight", C, bias=p + ".q_proj.bias", outT=Tq, group=grp, group_bwd=False)
        k = self.gemm(tok, p + ".k_proj.weight", C, bias=p + ".k_proj.bias", outT=Tk, group=grp, group_bwd=False)
        vv = self.gemm(tok, p + ".v_proj.weight", C, bias=p + ".v_proj.bias", outT=Tv, group=grp, group_bwd=False)
        self.group_end(grp)
        o, dOt = self.attention(q, k, vv, Tq["buf"], Tk["buf"], Tv["buf"], B, Hn, T, T)
        c = self.gemm(o, p + ".c_proj.weight", Cout, bias=p + ".c_proj.bias")
        self._patch_outT_on_dgrad(dOt)
        return self.bn(yc, stc, p + ".connect.1", relu=True, ident=c)

    def _patch_outT_on_dgrad(self, dOt):
        """The gemm layer appended last produced the attention output's consumer; in backward its dgrad must also
        write the head-split transposed copy dO^T that attn_bwd_dkv reads.  Wrap its closure accordingly."""
        if not self.training or dOt is None:
            return
        inner = self.tape.pop()
        eng = self

        def bwd():
            eng._dgrad_outT = dOt
            try:
                inner()
            finally:
                eng._dgrad_outT = None
        self.tape.append(bwd)

    # text --------------------------------------------------------------------------------------
    def _encode_text(self, word):
        B, L = word.shape
        D, Hn = self.clip.txt_width, self.clip.txt_heads
        x = Act(self.empty(B * L, D, dtype=F32), B, L, 1, D)
        ops.embed_fwd(word, self.P["backbone.token_embedding.weight"], self.P["backbone.positional_embedding"], x.t)
        x0 = x
        for i in range(self.clip.txt_layers):
            p = "backbone.transformer.resblocks.%d" % i
            h, _, _ = self.ln(x, p + ".ln_1", dx_stream=x)
            T3 = self.new_T(B, Hn, L, secs=3)
            qkv = self.gemm(h, p + ".attn.in_proj_weight", 3 * D, bias=p + ".attn.in_proj_bias", outT=T3)
            tb = T3["buf"]
            o, dOt = self.attention(qkv.slice(0, D), qkv.slice(D, D), qkv.slice(2 * D, D), tb[0], tb[1], tb[2], B, Hn, L, L, causal=True)
            x1 = self.new_act(B, L, 1, D, dtype=F32)
            self.gemm(o, p + ".attn.out_proj.weight", D, bias=p + ".attn.out_proj.bias", out=x1, resid=x, stream_grad=x1)
            self._patch_outT_on_dgrad(dOt)
            self._link_stream(x1, x)
            h2, _, _ = self.ln(x1, p + ".ln_2", dx_stream=x1)
            u = self.gemm(h2, p + ".mlp.c_fc.weight", 4 * D, bias=p + ".mlp.c_fc.bias")
            gl = self.new_act(B, L, 1, 4 * D)
            ops.quickgelu_fwd(u.t, gl.t)
            if self.training:
                def bwd_gelu(u=u, gl=gl):
                    du, acc = u.grad_target()
                    ops.quickgelu_bwd(u.t, gl.g, du)
                self.tape.append(bwd_gelu)
            x2 = self.new_act(B, L, 1, D, dtype=F32)
            self.gemm(gl, p + ".mlp.c_proj.weight", D, bias=p + ".mlp.c_proj.bias", out=x2, resid=x1, stream_grad=x2)
            self._link_stream(x2, x1)
            x = x2
        xf, _, _ = self.ln(x, "backbone.ln_final", dx_stream=x)
        eot = torch.empty(B, dtype=torch.int32, device=self.dev)
        if self.state_f32 and B <= ops.SMALL_MAX_ROWS:
            state = self._state_f32(word, x, xf, eot, B, L, D)
        else:
            rows = Act(self.empty(B, D), B, 1, 1, D)
            ops.eot_gather(word, xf.t, D, rows.t, eot)
            if self.training:
                def bwd_eot():
                    gx, acc = xf.grad_target()
                    if not acc:
                        ops.zero_(gx)
                    ops.eot_scatter_add(eot, rows.g, B, L, D, gx)
                self.tape.append(bwd_eot)
            state = self.gemm(rows, "backbone.text_projection", self.clip.embed_dim, w_transposed=True)
        if self.training:
            def bwd_embed():
                ops.embed_bwd(word, x0.g, self.G["backbone.token_embedding.weight"], self.G["backbone.positional_embedding"],
                              row_live=self.embed_live)
            self.tape.insert(self._text_tape_start, bwd_embed)
        return xf, state

    def _state_f32(self, word, x: Act, xf: Act, eot, B, L, D) -> Act:
        """state = LayerNorm(x)[eot] @ text_projection (model/clip.py:449-456) in fp32: the rows are normalised from the fp32
        residual stream with the statistics ln_final saved (not gathered from its bf16 output), the [B, D] x [D, E] product runs
        on the VALU from the fp32 parameter.  Returns an fp32 Act [B, E]; its consumers (neck.txt_proj, proj.txt) continue in fp32."""
        E = self.clip.embed_dim
        mean, rstd = xf.aux["ln_stats"]
        Pn = "backbone.text_projection"
        rows = torch.empty(B, D, dtype=F32, device=self.dev)
        ops.eot_gather_ln_f32(word, x.t, mean, rstd, self.P["backbone.ln_final.weight"], self.P["backbone.ln_final.bias"], D, rows, eot)
        state = Act(torch.empty(B, E, dtype=F32, device=self.dev), B, 1, 1, E)
        ops.linear_f32_small(rows, self.P[Pn], state.t, w_is_kn=True)
        if self.training:
            def bwd():
                ds = state.g                                                  # fp32 [B, E]: neck.txt_proj + proj.txt
                ops.outer_sum_f32_small(rows, ds, self.G[Pn])                 # dP[d][e] = sum_b rows[b][d] ds[b][e]
                drows = torch.empty(B, D, dtype=F32, device=self.dev)
                ops.linear_f32_small(ds, self.P[Pn], drows)                   # drows[b][d] = sum_e ds[b][e] P[d][e]
                gx, acc = xf.grad_target()
                if not acc:
                    ops.zero_(gx)
                ops.eot_scatter_add_f32(eot, drows, B, L, D, gx)
            self.tape.append(bwd)
        return state

    def _link_stream(self, new: Act, old: Act):
        """x_new = x_old + f(..): the stream gradient passes through unchanged - share one fp32 buffer."""
        if not self.training:
            return

        def bwd():
            if old.root._g is None:
                old.root._g = new.g
            elif old.root._g is not new.g:
                ops.axpy_f32(old.root._g, new.g, 1.0)
        # must run BEFORE (in backward order = appended AFTER) the branch closures that accumulate into old.g
        self.tape.append(bwd)

    # neck --------------------------------------------------------------------------------------
    def _fpn(self, v3: Act, v4: Act, v5: Act, state: Act) -> Act:
        n = "neck"
        fo = self.head.fpn_out
        B = v5.Bn
        # text projection: Linear(no bias) + BN1d + ReLU on [B, C]
        f32_head = state.t.dtype == F32
        if f32_head:
            s, s32, txt_bwd = None, *self._txt_proj_f32(state, n + ".txt_proj", fo[2])
        else:
            ys, sts = self.gemm(state, n + ".txt_proj.0.weight", fo[2], stats=True)
            s = self.bn(ys, sts, n + ".txt_proj.1")
            s32 = self.empty(B, fo[2], dtype=F32)
            ops.cast_bf16_f32(s.t, s32)
        y5, st5 = self.gemm(v5, n + ".f1_v_proj.0.weight", fo[2], stats=True)
        u = self.bn(y5, st5, n + ".f1_v_proj.1", mul=s32, want_stats=True)
        if self.training:
            if f32_head:
                bwd_s = lambda: txt_bwd(u.aux["dmul"])
            else:
                def bwd_s():
                    gs, acc = s.grad_target()
                    assert not acc
                    ops.cast_f32_bf16(u.aux["dmul"], gs)
            self.tape.insert(len(self.tape) - 1, bwd_s)          # runs after the bn(mul) backward that fills dmul
        f5 = self.bn(u, u.aux["stats"], n + ".norm_layer.0")
        # fusion 2: cat[f2_v_proj(v4), up2(f5)] -> f2_cat
        cat2 = self.new_act(v4.Bn, v4.H, v4.W, fo[1] + fo[2])
        self.conv_bn(v4, n + ".f2_v_proj.0", n + ".f2_v_proj.1", fo[1], k=3, pad=1, out=cat2.slice(0, fo[1]))
        self._upsample(f5, cat2.slice(fo[1], fo[2]))
        # fusion 3: cat[avgpool(f3_v_proj(v3)), f4] -> f3_cat   (f4 lives in its slice of the concat buffer)
        cat3 = self.new_act(v4.Bn, v4.H, v4.W, fo[0] + fo[1])
        f4 = self.conv_bn(cat2, n + ".f2_cat.0", n + ".f2_cat.1", fo[1], out=cat3.slice(fo[0], fo[1]))
        self.conv_bn(v3, n + ".f3_v_proj.0", n + ".f3_v_proj.1", fo[0], k=3, pad=1, pool=True, out=cat3.slice(0, fo[0]))
        f3 = self.conv_bn(cat3, n + ".f3_cat.0", n + ".f3_cat.1", fo[1])
        # fusion 4
        cat4 = self.new_act(v4.Bn, v4.H, v4.W, 3 * fo[1])
        # the three 3x3 convolutions of fusion 4 (model/layers.py:300-302) read f5 / f4 / f3 and write three buffers; their input
        # gradients go to three buffers as well: one grouped launch each way, see _group_variant
        G = self.group_begin()
        # (call r04b: forcing one tile on all three - 8w128x128 or 64x128 - is level with letting each keep its own: the two
        # M 5408 problems share a launch, the M 1352 one runs alone)
        gv = self._group_variant("f4_proj", "auto")
        y5p, s5p = self.gemm(f5, n + ".f4_proj5.0.weight", fo[1], k=3, pad=1, stats=True, group=G, variant=gv)
        y4p, s4p = self.gemm(f4, n + ".f4_proj4.0.weight", fo[1], k=3, pad=1, stats=True, group=G, variant=gv)
        y3p, s3p = self.gemm(f3, n + ".f4_proj3.0.weight", fo[1], k=3, pad=1, stats=True, group=G, variant=gv)
        self.group_end(G)
        fq5 = self.bn(y5p, s5p, n + ".f4_proj5.1")
        self.bn(y4p, s4p, n + ".f4_proj4.1", out=cat4.slice(fo[1], fo[1]))
        self.bn(y3p, s3p, n + ".f4_proj3.1", out=cat4.slice(0, fo[1]))
        self._upsample(fq5, cat4.slice(2 * fo[1], fo[1]))
        # aggregation + CoordConv (2 coordinate channels, zero padded to a multiple of 8)
        cc = self.new_act(v4.Bn, v4.H, v4.W, pad8(fo[1] + 2))
        self.conv_bn(cat4, n + ".aggr.0", n + ".aggr.1", fo[1], out=cc.slice(0, fo[1]))
        ops.fill_coords(cc.t, cc.ld, fo[1], cc.ld - fo[1], cc.Bn, cc.H, cc.W)
        self._neck_taps = dict(f5=f5, f4=f4, f3=f3, aggr=cc.slice(0, fo[1]), s=s if s is not None else s32)
        fq = self.conv_bn(cc, n + ".coordconv.0.conv1.0", n + ".coordconv.0.conv1.1", fo[1], k=3, pad=1)
        fq = self.conv_bn(fq, n + ".coordconv.1.0", n + ".coordconv.1.1", fo[1], k=3, pad=1)
        return fq

    def _txt_proj_f32(self, state: Act, p: str, C: int):
        """s = relu(BatchNorm1d(Linear(state))) (model/layers.py:262-264,286) over the B rows of the fp32 sentence vector: fp32
        kernels of csrc/smallf32.hip; the BatchNorm coefficients (and, with SyncBatchNorm, the exchange) come from the same
        launch as for every other BatchNorm.  Returns (s fp32 [B, C], backward(ds))."""
        B = state.Bn
        W = self.P[p + ".0.weight"]                                           # [C, E]
        pfx = p + ".1"
        ys = torch.empty(B, C, dtype=F32, device=self.dev)
        ops.linear_f32_small(state.t, W, ys)
        st = ops.colstats_f32_small(ys, self.dev) if self.training else None
        scale, shift, mean, invstd, gcount = self._bn_coeffs(pfx, st, float(B), C)
        s32 = torch.empty(B, C, dtype=F32, device=self.dev)
        ops.bn_relu_f32_small(ys, scale, shift, s32)

        def bwd(ds):
            Gb = self.G[pfx + ".bias"]
            arena_block = self.grad_arena[Gb.storage_offset():Gb.storage_offset() + 2 * C]       # [dbeta | dgamma]
            link = self.comm.next_link() if self.sync_bn else None
            if link is not None:
                sums = self.empty(2 * C, dtype=F32)
            elif self.sync_bn:
                sums = self.zeros(2 * C)
            else:
                sums = arena_block

            def between(t):
                ops.axpy_f32(arena_block, t, 1.0)
                ops.torch_op(self.comm.allreduce_sum_op(t))

            dys = torch.empty(B, C, dtype=F32, device=self.dev)
            ops.bn_relu_bwd_f32_small(ds, ys, scale, shift, mean, invstd, sums, gcount, dys,
                                      between=between if (self.sync_bn and link is None) else None, link=link,
                                      local_sums=arena_block if link is not None else None)
            ops.outer_sum_f32_small(dys, state.t, self.G[p + ".0.weight"])    # dW[c][e] = sum_b dys[b][c] state[b][e]
            gst, acc = state.grad_target()
            ops.linear_f32_small(dys, W, gst, w_is_kn=True, accumulate=acc)   # dstate[b][e] (+)= sum_c dys[b][c] W[c][e]

        return s32, bwd

    def _upsample(self, x: Act, out: Act):
        ops.upsample2_fwd(x.t, x.Bn, x.H, x.W, x.C, out.t, ldx=x.ld, xcoff=x.coff, ldy=out.ld, ycoff=out.coff)
        if self.training:
            def bwd():
                gx, acc = x.grad_target()
                ops.upsample2_bwd(out.g, x.Bn, x.H, x.W, x.C, gx, lddy=out.ld, dycoff=out.coff, lddx=x.ld, dxcoff=x.coff, accum=acc)
            self.tape.append(bwd)

    def _copy(self, x: Act, out: Act):
        ops.add_bf16(x.t, out.t, x.M, x.C, lda=x.ld, acoff=x.coff, ldy=out.ld, ycoff=out.coff)
        if self.training:
            def bwd():
                gx, acc = x.grad_target()
                ops.add_bf16(out.g, gx, x.M, x.C, b=gx if acc else None, lda=out.ld, acoff=out.coff, ldb=x.ld, bcoff=x.coff,
                             ldy=x.ld, ycoff=x.coff)
            self.tape.append(bwd)

    # decoder -----------------------------------------------------------------------------------
    def _decoder(self, fq: Act, txt: Act, word) -> Act:
        hd = self.head
        B, H, W, C = fq.Bn, fq.H, fq.W, fq.C
        HW, L, Hn = H * W, txt.H, hd.num_head
        vpos = self.table(("pos2d", C, H, W), lambda: tables.pos2d_table(C, H, W))
        tpos = self.table(("pos1d", txt.C, L), lambda: tables.pos1d_table(txt.C, L))
        vis = self.new_act(B, H, W, C, dtype=F32)
        ops.cast_bf16_f32(fq.t, vis.t)
        vis0 = vis
        txtpos = self.new_act(txt.Bn, L, 1, txt.C)
        ops.add_rowtable(txt.t, tpos, L, txtpos.t, txt.M, txt.C)
        if self.training:
            def bwd_txtpos():
                if txtpos.g is None:
                    return
                gx, acc = txt.grad_target()
                ops.add_bf16(txtpos.g, gx, txt.M, txt.C, b=gx if acc else None)
            self.tape.append(bwd_txtpos)
        for i in range(hd.num_layers):
            p = "decoder.layers.%d" % i
            # --- self attention ---
            v2, qk, _ = self.ln(vis, p + ".norm1", pos=vpos, pos_rows=HW, dx_stream=vis)
            T2, Tv = self.new_T(B, Hn, HW, secs=2), self.new_T(B, Hn, HW)
            # the q|k projection (of norm1(vis) + pos) and the v projection (of norm1(vis)) are independent, forward and backward
            # (model/layers.py:202-207): one grouped launch each way
            G = self.group_begin()
            qkp = self.gemm(qk, p + ".self_attn.in_proj_weight", 2 * C, rows=(0, 2 * C), bias=p + ".self_attn.in_proj_bias", outT=T2, group=G)
            vp = self.gemm(v2, p + ".self_attn.in_proj_weight", C, rows=(2 * C, 3 * C), bias=p + ".self_attn.in_proj_bias", outT=Tv, group=G)
            self.group_end(G)
            o, dOt = self.attention(qkp.slice(0, C), qkp.slice(C, C), vp, T2["buf"][0], T2["buf"][1], Tv["buf"][0], B, Hn, HW, HW,
                                    drop=self.drop(i, 0))
            a = self.gemm(o, p + ".self_attn.out_proj.weight", C, bias=p + ".self_attn.out_proj.bias")
            self._patch_outT_on_dgrad(dOt)
            _, _, vis1 = self.ln(a, p + ".self_attn_norm", want_y=False, resid=vis, out_drop=self.drop(i, 1))
            # --- cross attention ---
            _, qc, _ = self.ln(vis1, p + ".norm2", want_y=False, pos=vpos, pos_rows=HW, dx_stream=vis1)
            Tq, Tk, Tvv = self.new_T(B, Hn, HW), self.new_T(B, Hn, L), self.new_T(B, Hn, L)
            m = p + ".multihead_attn"
            # cross attention: q of the pixels, k / v of the words (model/layers.py:235-243) - three inputs, three gradient buffers
            G = self.group_begin()
            qx = self.gemm(qc, m + ".in_proj_weight", C, rows=(0, C), bias=m + ".in_proj_bias", outT=Tq, group=G)
            kx = self.gemm(txtpos, m + ".in_proj_weight", C, rows=(C, 2 * C), bias=m + ".in_proj_bias", outT=Tk, group=G)
            vx = self.gemm(txt, m + ".in_proj_weight", C, rows=(2 * C, 3 * C), bias=m + ".in_proj_bias", outT=Tvv, group=G)
            self.group_end(G)
            o2, dOt2 = self.attention(qx, kx, vx, Tq["buf"][0], Tk["buf"][0], Tvv["buf"][0], B, Hn, HW, L, key_tokens=word,
                                      drop=self.drop(i, 2))
            a2 = self.gemm(o2, m + ".out_proj.weight", C, bias=m + ".out_proj.bias")
            self._patch_outT_on_dgrad(dOt2)
            _, _, vis2 = self.ln(a2, p + ".cross_attn_norm", want_y=False, resid=vis1, out_drop=self.drop(i, 3))
            # --- FFN: Linear -> ReLU -> Dropout -> LayerNorm -> Linear, + dropout3 residual ---
            h3, _, _ = self.ln(vis2, p + ".norm3", dx_stream=vis2)
            u = self.gemm(h3, p + ".ffn.0.weight", hd.dim_ffn, bias=p + ".ffn.0.bias")
            hn, _, _ = self.ln(u, p + ".ffn.3", in_relu=True, in_drop=self.drop(i, 4))
            vis3 = self.new_act(B, H, W, C, dtype=F32)
            self.gemm(hn, p + ".ffn.4.weight", C, bias=p + ".ffn.4.bias", out=vis3, resid=vis2, drop=self.drop(i, 5), stream_grad=vis3)
            self._link_stream(vis3, vis2)
            vis = vis3
        out, _, _ = self.ln(vis, "decoder.norm", dx_stream=vis)
        if self.training:
            def bwd_in():
                gx, acc = fq.grad_target()
                assert not acc
                ops.cast_f32_bf16(vis0.g, gx)
            self.tape.insert(self._dec_tape_start, bwd_in)
        return out

    # projector + loss --------------------------------------------------------------------------
    def _projector(self, fq: Act, state: Act):
        p = "proj"
        c = self.head.vis_dim // 2
        B = fq.Bn
        u1 = self.new_act(B, fq.H * 2, fq.W * 2, fq.C)
        self._upsample(fq, u1)
        z1 = self.conv_bn(u1, p + ".vis.1.0", p + ".vis.1.1", 2 * c, k=3, pad=1)
        u2 = self.new_act(B, z1.H * 2, z1.W * 2, z1.C)
        self._upsample(z1, u2)
        z2 = self.conv_bn(u2, p + ".vis.3.0", p + ".vis.3.1", c, k=3, pad=1)
        x = self.gemm(z2, p + ".vis.4.weight", c, bias=p + ".vis.4.bias")
        nwb = c * 9 + 1
        wb = Act(self.zeros(B, pad8(nwb)), B, 1, 1, nwb, pad8(nwb))
        self._wb_layer(state, wb, nwb)
        pred = self.empty(B, 1, x.H, x.W, dtype=F32)
        ops.dynconv_fwd(x.t, B, x.H, x.W, c, wb.t, pred)
        return pred, x, wb

    def _wb_layer(self, state: Act, wb: Act, nwb: int):
        """wb = Linear(state) in fp32 (per-sample 3x3 kernel + bias of the text-to-pixel conv)."""
        p = "proj.txt"
        if state.t.dtype == F32:
            W = self.P[p + ".weight"]                                         # [nwb, E]
            ops.linear_f32_small(state.t, W, wb.t[:, :nwb], bias=self.P[p + ".bias"])
            if self.training:
                def bwd32():
                    dwb = self._dwb[:, :nwb]                                  # fp32 [B, nwb] filled by dynconv_bwd
                    ops.outer_sum_f32_small(dwb, state.t, self.G[p + ".weight"], rowsum=self.G[p + ".bias"])
                    gst, acc = state.grad_target()
                    ops.linear_f32_small(dwb, W, gst, w_is_kn=True, accumulate=acc)
                self.tape.append(bwd32)
            return
        g = Geom.linear(state.M, state.C)
        ops.conv_gemm(state.t, self.WF[p + ".weight"], g, nwb, lda=state.ld, a_coff=state.coff, bias=self.P[p + ".bias"],
                      out=wb.t, ldc=wb.ld)
        if self.training:
            def bwd():
                dwb = self._dwb                                        # fp32 [B, ld] filled by dynconv_bwd
                gy = self.empty(state.M, wb.ld)
                ops.cast_f32_bf16(dwb, gy)
                ops.conv_wgrad(gy, state.t, g, nwb, self.G[p + ".weight"], ldy=wb.ld, N_ld=wb.ld, ldx=state.ld, x_coff=state.coff,
                               dbias=self.G[p + ".bias"], queue=self._wq)
                gx, acc = state.grad_target()
                gD = Geom.linear(state.M, wb.ld)
                ops.conv_gemm(gy, self.WD[p + ".weight"], gD, state.C, lda=wb.ld, out=gx, ldc=state.ld, c_coff=state.coff,
                              resid=gx if acc else None, ldr=state.ld, r_coff=state.coff)
            self.tape.append(bwd)

    # ------------------------------------------------------------------------------------------
    # whole step
    # ------------------------------------------------------------------------------------------
    def forward(self, img, word, mask=None, training=True, seed=0, taps: Optional[dict] = None):
        self.training, self.seed = training, int(seed) & 0xFFFFFFFF
        self.tape = []
        self._wq, self._sq = ops.WgradQueue(self._flush_wgrads), ops.SumQueue()      # (drops what a forward without backward left queued)
        self._dgrad_outT = None
        self._stage_marks = {}
        Act._engine = self
        self._zero_slab_begin()
        if training:
            self.comm.begin_step()
        if not self.packs_current:
            self.repack_weights()
        # token ids index the embedding table and the key-padding mask as int64 (torch.nn.Embedding would raise on anything
        # else); the expression may not be longer than the text positional table (reference model/clip.py:441)
        if word.dtype != torch.int64:
            if word.is_floating_point():
                raise TypeError("word must hold integer token ids, got %s" % word.dtype)
            word = word.to(torch.int64)
        word = word.contiguous()
        if word.shape[1] > self.P["backbone.positional_embedding"].shape[0]:
            raise ValueError("expression length %d exceeds the text context length %d"
                             % (word.shape[1], self.P["backbone.positional_embedding"].shape[0]))
        main = torch.cuda.current_stream()
        self._text_tape_start = 0
        self._vis_stage_start = {}
        if self.side is not None:
            ops.torch_op(lambda: self.side.wait_stream(main))
            with torch.cuda.stream(self.side):
                txt, state = self._encode_text(word)
        else:
            txt, state = self._encode_text(word)
        v0 = len(self.tape)
        self._vis_stage_start[0] = v0
        v3, v4, v5, feats = self._encode_image(img.contiguous().float())
        if self.side is not None:
            ops.torch_op(lambda: main.wait_stream(self.side))
        v1 = len(self.tape)
        self._ranges = dict(text=(0, v0), visual=(v0, v1))
        head_marks = {5: v1}
        fq = self._fpn(v3, v4, v5, state)
        self._dec_tape_start = head_marks[6] = len(self.tape)
        fqd = self._decoder(fq, Act(txt.t, txt.Bn, txt.H, 1, txt.C, root=txt.root), word)
        head_marks[7] = len(self.tape)
        pred, x, wb = self._projector(fqd, state)
        # a stage's parameter gradients are complete once the closure at its start index has run in backward
        self._stage_marks = {}
        for st, idx in list(head_marks.items()) + list(self._vis_stage_start.items()):
            self._stage_marks.setdefault(idx, []).append(st)
        if taps is not None:
            taps.update(self._neck_taps)
            taps.update(layer1=feats[0], layer2=feats[1], layer3=feats[2], layer4=feats[3], attnpool=v5, word=txt, state=state,
                        fq_neck=fq, fq_dec=fqd, pred=pred)
        if not training:
            Act._engine = None
            return pred
        B, _, OH, OW = pred.shape
        msk = self.empty(B, 1, OH, OW, dtype=F32)
        ops.mask_resize_nearest(mask.contiguous().float(), OH, OW, msk)
        loss = self.empty(1, dtype=F32)
        ops.bce_fwd(pred, msk, loss)
        c = self.head.vis_dim // 2

        def bwd_loss():
            dpred = self.empty(B, 1, OH, OW, dtype=F32)
            ops.bce_bwd(pred, msk, self._gscale, dpred)
            gx, acc = x.grad_target()
            assert not acc
            self._dwb = self.zeros(B, wb.ld)
            ops.dynconv_bwd(x.t, dpred, B, OH, OW, c, wb.t, gx, self._dwb)

        # the loss closure must run first, then the wb layer (needs _dwb) - it was appended before the dynconv
        self.tape.append(bwd_loss)
        return pred, msk, loss.view(())

    def backward(self, gscale: Optional[torch.Tensor] = None, on_stage_done: Optional[Callable[[int], None]] = None):
        """Run the tape in reverse.  `gscale`: optional 1-element fp32 device tensor multiplying dloss (GradScaler).
        `on_stage_done(stage)` fires when every gradient of an arena stage (7 projector .. 0 stem+layer1, see
        _build_grad_arena) has been issued on the current stream - the hook the data-parallel gradient exchange uses to
        overlap with the rest of backward."""
        self._gscale = gscale
        Act._engine = self                              # (another engine may have run a forward in between)
        # the accumulated parts of the gradient arena (BatchNorm sums, embedding rows) are cleared HERE, not in forward: a forward
        # never touches the arena, so gradients a caller still holds from the previous step survive it (drop-in module)
        if debug.HOOKS.zero_all:
            ops.zero_(self.grad_arena)
        else:
            ops.zero_ranges(self.zero_ranges)
        marks = dict(self._stage_marks)                 # tape index at which a stage's closures START
        (t0, t1), (v0, v1) = self._ranges["text"], self._ranges["visual"]
        head_start = max(t1, v1)                        # neck / decoder / projector closures: main stream
        def fire(i):
            if i in marks:
                self._flush_queues()                         # the stage's queued weight gradients (current stream)
                if on_stage_done is not None:
                    for st in marks[i]:
                        on_stage_done(st)

        for i in range(len(self.tape) - 1, head_start - 1, -1):
            self.tape[i]()
            fire(i)
        self._flush_queues()
        if debug.HOOKS.hold_backward_fork:
            torch.cuda._sleep(60000000)
        # the two encoders' backward passes are independent: text on the side stream, visual on the launch stream
        main = torch.cuda.current_stream()

        def text_section():
            for i in range(t1 - 1, t0 - 1, -1):
                self.tape[i]()
                if debug.HOOKS.taps is not None:
                    debug.tap(i, self.tape[i])
            self._flush_queues()
            if on_stage_done is not None:
                on_stage_done(4)                         # issued from the side stream: the exchange waits for it only

        def visual_section():
            for i in range(v1 - 1, v0 - 1, -1):
                self.tape[i]()
                fire(i)                                  # visual stages 3, 2, 1, 0 as their layer groups finish
            self._flush_queues()

        if self.side is not None:
            ops.torch_op(lambda: self.side.wait_stream(main))
            with torch.cuda.stream(self.side):
                text_section()
            visual_section()
            ops.torch_op(lambda: main.wait_stream(self.side))
        else:
            text_section()
            visual_section()
        self._zneed_last = max(self._zneed_last, self._zneed)
        Act._engine = None
        self.tape = []
        return self.G
