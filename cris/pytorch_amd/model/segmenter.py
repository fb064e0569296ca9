"""`CRIS` - drop-in for the reference's `model.segmenter.CRIS` (reference model/segmenter.py:10-62) on the HIP path.

Same constructor (`CRIS(cfg)` reading clip_pretrain, word_len, fpn_in, fpn_out, num_layers, vis_dim, num_head, dim_ffn,
dropout, intermediate, word_dim), same `forward(img, word, mask=None)` contract - train mode returns
`(pred.detach(), mask_resized, loss)`, eval mode `pred.detach()` - same parameter / buffer names and shapes, so the
reference's `train.py` / `engine/engine.py` drive it unchanged: `.cuda()`, `.train()/.eval()`, `state_dict()` /
`load_state_dict()` (checkpoint interchange), `nn.SyncBatchNorm.convert_sync_batchnorm`, `DistributedDataParallel(...,
find_unused_parameters=True)`, ambient `torch.cuda.amp.autocast()` + `GradScaler`, any `torch.optim` optimizer.

The module's children are parameter HOLDERS (arch.build_param_tree); every FLOP runs in libcris_hip.so through
engine.Engine.  Autograd sees ONE node (`_CrisStep`): its forward runs the whole HIP forward + loss, its backward runs the
HIP backward tape and hands autograd one gradient per parameter (3x3-conv gradients are converted from the GEMM layout
by `cris_unpack_grads`), so DDP's bucketed all-reduce, GradScaler's unscale / inf check and the optimizer see ordinary
`.grad` tensors.  There is no eager / CPU fallback: without the HIP library or off the GPU, forward raises.
"""
import os
import weakref

import torch
import torch.distributed as dist
from torch import nn

from .. import arch, debug, ops
from ..engine import Engine

_FP16_ROUNDED_SUFFIXES = ("in_proj_weight", "in_proj_bias", "q_proj_weight", "k_proj_weight", "v_proj_weight")


def _cfg_get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def head_spec_from_cfg(cfg) -> arch.HeadSpec:
    return arch.HeadSpec(word_len=int(_cfg_get(cfg, "word_len")), fpn_in=tuple(_cfg_get(cfg, "fpn_in")),
                         fpn_out=tuple(_cfg_get(cfg, "fpn_out")), num_layers=int(_cfg_get(cfg, "num_layers")),
                         vis_dim=int(_cfg_get(cfg, "vis_dim")), num_head=int(_cfg_get(cfg, "num_head")),
                         dim_ffn=int(_cfg_get(cfg, "dim_ffn")), dropout=float(_cfg_get(cfg, "dropout")),
                         intermediate=bool(_cfg_get(cfg, "intermediate", False)), word_dim=int(_cfg_get(cfg, "word_dim")))


def load_clip_state_dict(path):
    """The CLIP archive's state_dict (reference model/segmenter.py:14-15: torch.jit.load(cfg.clip_pretrain).state_dict()).
    `synthetic`, `synthetic:r101`, `synthetic:tiny` generate the deterministic random archive of arch.py instead (there is
    no network for pretrain/RN50.pt in this environment)."""
    if isinstance(path, str) and path.startswith("synthetic"):
        name = path.split(":", 1)[1] if ":" in path else "r50"
        clip, head = arch.specs_by_name(name)
        return arch.clip_state_dict_view(arch.synthetic_state_dict(clip, head, 0))
    if not os.path.isfile(path):
        raise FileNotFoundError("CLIP archive %r not found (cfg.clip_pretrain)" % (path,))
    return torch.jit.load(path, map_location="cpu").eval().state_dict()


def _is_clip_half_tensor(name, holder):
    """Which CLIP tensors the reference's loader rounds to fp16 before `.float()` (model/clip.py:477-500,552;
    model/segmenter.py:16): Conv / Linear weights and biases, MultiheadAttention projections, text_projection."""
    if name == "text_projection" or name.endswith(_FP16_ROUNDED_SUFFIXES):
        return True
    parent = name.rsplit(".", 1)[0] if "." in name else ""
    try:
        m = holder.get_submodule(parent)
    except AttributeError:
        return False
    return isinstance(m, (nn.Conv2d, nn.Linear))


class _CrisStep(torch.autograd.Function):
    """One autograd node for the whole HIP forward + loss; backward = the engine's tape."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, module, img, word, mask, seed, *params):
        eng = module._engine
        ctx.module = module
        ctx.direct = len(params) == 1 and params[0] is module._anchor
        # SURVEY 8e option B: under a DistributedDataParallel wrapper that was told to ignore this module's parameters the
        # gradient exchange is the module's own (the communicator, or None: no exchange in this step)
        ctx.exchange = module._exchange_comm_for_this_step()
        ctx.step_id = module._steps                 # the engine keeps ONE step's saved state (tape / captured buffers)
        st = module._graph_step(img, word, mask, seed, ctx.exchange)
        ctx.graph = st
        if st is not None:                       # replayed HIP graph: outputs are the capture's static buffers (fresh aliases)
            pred, msk, loss = st["pred"].view_as(st["pred"]), st["msk"].view_as(st["msk"]), st["loss"].view(())
        else:
            pred, msk, loss = module._training_forward(img, word, mask, seed)
        ctx.mark_non_differentiable(pred, msk)
        return pred, msk, loss

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, _gpred, _gmsk, gloss):
        module = ctx.module
        eng = module._engine
        if ctx.step_id != module._steps:
            raise RuntimeError("CRIS (HIP path): backward of a training forward whose saved state a NEWER training forward has "
                               "overwritten - the engine keeps the activations of one step (call backward before the next "
                               "training forward; losses of several forwards cannot be back-propagated together)")
        gscale = gloss.detach().reshape(1).to(torch.float32).contiguous()        # GradScaler's factor arrives here
        xchg = ctx.exchange
        if xchg is not None:
            # DistributedDataParallel divides every gradient by the world size BEFORE it adds them up over the ranks
            # (reducer: bucket = grad / world, all-reduce SUM): the same order here, through the factor on dloss
            gscale = gscale * (1.0 / xchg.world)
        if ctx.direct:
            module._release_engine_grads()       # before the backward pass overwrites the engine's gradient buffers
        if xchg is not None:
            module._exchange_stale_grads(xchg)   # gradients of no_sync() micro-batches still sitting in `.grad`
        st = ctx.graph
        if st is not None and st.get("bwd") is None:
            # command-list mode, first step with this shape: the forward was recorded while it ran, now the backward is
            st["gscale"].copy_(gscale, non_blocking=True)
            module._record_backward(st, xchg)
            grads = st["grads"]
        elif st is not None:
            st["gscale"].copy_(gscale, non_blocking=True)
            st["bwd"].replay()
            grads = st["grads"]
        else:
            module._backward_and_exchange(gscale, xchg)
            grads = module._export_grads()
        if ctx.direct:
            # single process, no per-parameter hooks: the gradients ARE the engine's buffers - hand them to `.grad` directly
            # instead of letting 449 AccumulateGrad nodes clone them one by one (1.8 ms of launches per R50 step)
            module._assign_grads(grads)
            return (None, None, None, None, None, None)
        return (None, None, None, None, None) + tuple(grads)


# Bumped whenever a module that belongs to a live CRIS tree registers a parameter or a sub-module (torch's global registration
# hooks; assignments through __setattr__ go through them too): CRIS._fast_key re-walks its parameter list when this moved.
# Round-4 advisor finding: the fast key looked at the first and last parameter only, so a replaced MIDDLE parameter kept training
# the old tensor.  Round-5 finding: the hooks are process-global, and a loop that builds ANY small nn.Module per step (a loss, a
# metric with a buffer-holding child) bumped the epoch and forced the ~1.5 ms tree walk on every forward - so only modules of a
# live CRIS tree count (`_TREE_MODULES`: every sub-module seen by the last walk of each live CRIS; a module that is being
# CONSTRUCTED is not in it yet, and a module grafted into a tree is registered by a member of the tree, which is).
_REGISTRATION_EPOCH = [0]
_TREE_MODULES = weakref.WeakSet()


def _bump_registration_epoch(module, *_a, **_k):
    if module in _TREE_MODULES:
        _REGISTRATION_EPOCH[0] += 1
    return None


nn.modules.module.register_module_parameter_registration_hook(_bump_registration_epoch)
nn.modules.module.register_module_module_registration_hook(_bump_registration_epoch)
_data_ptr = torch.Tensor.data_ptr
_PEER_CHECK_EVERY = int(os.environ.get("CRIS_PEER_CHECK_EVERY", "200"))


class CRIS(nn.Module):
    _instances = weakref.WeakSet()          # live modules (cris.pytorch_amd.optim.Adam looks its parameters' owner up here)

    def __init__(self, cfg):
        super().__init__()
        CRIS._instances.add(self)
        self._grad_views = False            # set by cris.pytorch_amd.optim.Adam: `.grad` = views of the gradient arena, no conversion pass
        self._grad_views_active = False
        clip_sd = load_clip_state_dict(_cfg_get(cfg, "clip_pretrain"))
        self.clip_spec = arch.clip_spec_from_state_dict(clip_sd)
        self.head_spec = head_spec_from_cfg(cfg)
        tree = arch.build_param_tree(self.clip_spec, self.head_spec)
        # Vision & Text Encoder (model/segmenter.py:13-16), Multi-Modal FPN (:18), Decoder (:20-25), Projector (:27)
        self.backbone, self.neck, self.decoder, self.proj = tree.backbone, tree.neck, tree.decoder, tree.proj
        own = self.backbone.state_dict()
        load = {}
        for k, v in clip_sd.items():
            if k in ("input_resolution", "context_length", "vocab_size"):       # dropped by the reference (clip.py:548-550)
                continue
            if k not in own:
                continue                                                        # strict=False (attnpool.connect.* is new)
            t = v.detach().float()
            if _is_clip_half_tensor(k, self.backbone):
                t = t.half().float()
            load[k] = t
        self.backbone.load_state_dict(load, strict=False)
        self._engine = None
        self._engine_key = None
        self.graph_error = None
        self._steps = 0
        self._unpack = None
        self._ddp_asked = False             # a DistributedDataParallel constructor read _ddp_params_and_buffers_to_ignore
        self._ddp_ref = None                # weak reference to that wrapper (its no_sync() state is honoured)
        self._ddp_extra_ignore = []         # names a caller set through DDP's _set_params_and_buffers_to_ignore_for_model
        self._self_exchange = False         # decided per engine (_ensure_engine)
        self._ddp_synced = False
        self._xgen_dev, self.syncbn_exchange = None, "none"
        self.grad_exchange = "rccl"

    # ------------------------------------------------------------------------------------------------
    # SURVEY.md 8e option B - the gradient exchange under the reference's DistributedDataParallel wrap (train.py:100-102).
    # DDP reads `module._ddp_params_and_buffers_to_ignore` in its constructor: every parameter named there gets no
    # AccumulateGrad hook, no bucket, no initial broadcast.  The module names ALL of its parameters but `backbone.logit_scale`
    # (DDP refuses a module without a parameter that requires a gradient; that one receives none in the reference either) and
    # then does what DDP would have done, on the library's own terms: rank 0's parameters and buffers are broadcast once, each
    # backward all-reduces the flat gradient arena as eight stage-sized messages on a side stream and a communicator of their
    # own while the earlier stages are still in backward (the exchange NativeTrainer uses), divided by the world size before
    # the sum as DDP's reducer does, and `.grad` is handed out directly.  What this removes per step: 449 AccumulateGrad
    # clones, DDP's bucket copies in and out, its per-forward buffer broadcast - ~7 ms of 22 at R50 / 416 / batch 8 (round 5).
    # CRIS_DDP_SELF_EXCHANGE=0 hides the attribute: DDP then manages every parameter itself, as in rounds 2-5.
    @property
    def _ddp_params_and_buffers_to_ignore(self):
        if os.environ.get("CRIS_DDP_SELF_EXCHANGE", "1") != "1":
            raise AttributeError("_ddp_params_and_buffers_to_ignore")
        try:
            import sys
            from torch.nn.parallel import DistributedDataParallel as DDP
            w = sys._getframe(1).f_locals.get("self")
            if isinstance(w, DDP):
                self._ddp_asked = True
                self._ddp_ref = weakref.ref(w)
        except Exception:                        # noqa: BLE001 - no frame introspection: believe that it was a DDP constructor
            self._ddp_asked = True
        names = [n for n, _ in self.named_parameters() if n != "backbone.logit_scale"]
        # BatchNorm buffers: with SyncBatchNorm every rank computes the same running statistics, DDP's per-forward broadcast of
        # rank 0's would be 213 tensors of traffic for nothing; with plain BatchNorm under DDP the reference DOES overwrite the
        # other ranks' statistics with rank 0's in every forward (broadcast_buffers=True) - those stay DDP's to manage
        if isinstance(self.backbone.visual.bn1, nn.SyncBatchNorm):
            names += [n for n, _ in self.named_buffers()]
        return names + list(self._ddp_extra_ignore)

    @_ddp_params_and_buffers_to_ignore.setter
    def _ddp_params_and_buffers_to_ignore(self, value):
        self._ddp_extra_ignore = list(value)

    def _exchange_comm_for_this_step(self):
        """the communicator this training step's backward exchanges gradients on, or None (not under a DDP wrapper that left
        the parameters to the module / one rank / inside the wrapper's no_sync())"""
        if not self._self_exchange:
            return None
        comm = self._engine.comm
        if comm.world <= 1 and not debug.HOOKS.force_dist:
            return None
        ddp = self._ddp_ref() if self._ddp_ref is not None else None
        if ddp is not None and not ddp.require_backward_grad_sync:
            return None
        return comm

    def _training_forward(self, img, word, mask, seed):
        """the engine's training forward; with the SyncBN mailboxes the exchange generation advances first (a launch of its own,
        so that a captured / recorded forward advances it on every replay)"""
        if self._xgen_dev is not None:
            ops.counter_advance(self._xgen_dev)
        return self._engine.forward(img, word, mask, training=True, seed=seed)

    def _check_peers(self):
        """COLLECTIVE, every CRIS_PEER_CHECK_EVERY-th training forward (default 200; 0 = never): raise on every rank when a
        mailbox exchange of any rank gave up waiting for a peer (dist.TorchDistComm.check_peer_timeout) - a rank that lost a
        peer must not train on alone on poisoned statistics (round-5 advisor finding: nothing ever called the check)."""
        every = _PEER_CHECK_EVERY
        if every > 0 and self._xgen_dev is not None and self._steps % every == 0 and self._steps > 0:
            chk = getattr(self._engine.comm, "check_peer_timeout", None)
            if chk is not None:
                chk()

    def _backward_and_exchange(self, gscale, comm):
        """the engine's backward; with a communicator each arena stage is all-reduced (SUM) on the communicator's side stream as
        soon as backward has issued its last gradient, and the launch stream joins the exchange at the end"""
        eng = self._engine
        if comm is None:
            eng.backward(gscale=gscale)
            return

        def on_stage(st):
            lo, hi = eng.stage_ranges[st]
            ops.torch_op(lambda: comm.allreduce_async(eng.grad_arena[lo:hi]))
        eng.backward(gscale=gscale, on_stage_done=on_stage)
        ops.torch_op(comm.wait_all)

    def _exchange_stale_grads(self, comm):
        """Gradients that are still in `.grad` when an exchanging backward starts were accumulated by backward passes inside the
        wrapper's no_sync() (local, not averaged): DDP would average old + new together; here the old part is averaged now
        (tensor by tensor: the slow form, for a case the reference's loop never enters - it zeroes with set_to_none) and the
        new part arrives averaged.  Values that were averaged already (identical on every rank) are unchanged by this."""
        stale = [p.grad for p in self._step_params if p.grad is not None]
        for g in stale:
            g.div_(comm.world)
            comm.allreduce_sum(g)

    def _sync_from_rank0(self):
        """what DistributedDataParallel's constructor does for the parameters it manages (_sync_module_states): every rank
        starts from rank 0's parameters and buffers (the decoder / neck / projector are randomly initialised per process)"""
        for t in list(self.parameters()) + [b for b in self.buffers() if b.is_floating_point() or b.dtype == torch.int64]:
            dist.broadcast(t.data, 0)

    # ------------------------------------------------------------------------------------------------
    def _grad_params(self):
        return [(n, p) for n, p in self.named_parameters() if n != "backbone.logit_scale"]

    def _apply(self, fn, *args, **kwargs):
        self._plist = None                   # .cuda() / .to() / .float(): the parameter tensors may move
        return super()._apply(fn, *args, **kwargs)

    def _fast_key(self, device):
        """what can change the engine between two forwards, without walking the module tree (two traversals of the 600
        submodules cost ~1.5 ms of host time per step - time in which the GPU idles under the reference's loop): the device, the
        storage of the first and last parameter (a move re-allocates all of them; in-place loads keep them), the kind of the
        BatchNorm modules (convert_sync_batchnorm replaces them all), whether a process group exists, the gradient mode"""
        if getattr(self, "_plist", None) is None or self._plist_epoch != _REGISTRATION_EPOCH[0]:
            # (a Parameter or sub-module registered ANYWHERE since the list was made - weight surgery, a re-initialised head -
            # makes the cached list suspect: walk the tree again, once)
            self._plist = list(self.parameters())
            self._plist_epoch = _REGISTRATION_EPOCH[0]
            for m in self.modules():
                _TREE_MODULES.add(m)
        pl = self._plist
        # every parameter's storage address (`p.data = other` re-points one tensor without registering anything): ~40 us of
        # host time for the 449 tensors, against ~1.5 ms for the tree walk
        return (str(device), sum(map(_data_ptr, pl)), pl[0].data_ptr(), pl[-1].data_ptr(), len(pl), type(self.backbone.visual.bn1),
                dist.is_available() and dist.is_initialized(), bool(self._grad_views), self._ddp_asked)

    def _ensure_engine(self, device):
        fk = self._fast_key(device)
        if self._engine is not None and fk == getattr(self, "_engine_fast_key", None):
            return self._engine
        self._engine_fast_key = fk
        params = {n: p.data for n, p in self.named_parameters()}
        buffers = {n: b for n, b in self.named_buffers() if n.endswith(("running_mean", "running_var"))}
        # (CRIS_FORCE_DIST=1: the multi-rank code paths with a world of one - what one GPU can run of them under RCCL)
        sync = (any(isinstance(m, nn.SyncBatchNorm) for m in self.modules()) and dist.is_available()
                and dist.is_initialized() and (dist.get_world_size() > 1 or debug.HOOKS.force_dist))
        # gradient-view mode (cris.pytorch_amd.optim.Adam bound): also under a process group / DistributedDataParallel - there
        # `.grad` is DDP's averaged gradient in its own tensor, which the optimizer copies into the arena views before its
        # fused update (optim.Adam.step)
        views = bool(self._grad_views)
        pg = dist.is_available() and dist.is_initialized()
        xchg = bool(self._ddp_asked and pg)
        key = (str(device), sync, views, xchg, tuple(p.data_ptr() for p in params.values()), tuple(b.data_ptr() for b in buffers.values()))
        if self._engine is not None and key == self._engine_key:
            return self._engine
        self._self_exchange = xchg
        if xchg and dist.get_world_size() > 1 and not self._ddp_synced:
            self._sync_from_rank0()
            self._ddp_synced = True
        for n, p in params.items():
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("CRIS (HIP path) needs contiguous fp32 parameters, got %s %s" % (n, p.dtype))
        comm = None
        if sync or (xchg and (dist.get_world_size() > 1 or debug.HOOKS.force_dist)):
            from ..dist import TorchDistComm
            comm = TorchDistComm(device)
        self._engine = Engine(self.clip_spec, self.head_spec, params, buffers, device, comm=comm, sync_bn=sync)
        self._engine_key = key
        # SyncBN statistics (142 exchanges of a few KB per step, all on the critical path): through the peer-mapped mailboxes,
        # inside the BatchNorm launches, as under NativeTrainer - when allocation, IPC mapping and a self-test with known data
        # succeed on EVERY rank; otherwise (and with CRIS_SYNCBN_P2P=0) one RCCL all-reduce per exchange.  Round 6: until now
        # the drop-in module always took the collectives - ~20 us each over xGMI, ~3 ms of every 8-GPU step.
        self._xgen_dev, self.syncbn_exchange = None, "none" if not self._engine.sync_bn else "collective"
        if (self._engine.sync_bn and comm is not None and os.environ.get("CRIS_SYNCBN_P2P", "1") == "1"
                and torch.device(device).type == "cuda" and hasattr(comm, "enable_p2p")):
            e_ = self._engine
            self._xgen_dev = torch.zeros(1, dtype=torch.int32, device=device)       # generation of the exchanges: +1 per training forward, never rewound
            cmax = max(e_.P[pfx + ".weight"].numel() for pfx in e_.bn_prefixes)
            why = comm.enable_p2p(slots=2 * len(e_.bn_prefixes) + 8, max_floats=4 * cmax, gen_dev=self._xgen_dev)
            if why is None:
                self.syncbn_exchange = ("p2p mailboxes, exchanged inside the BatchNorm launches" if getattr(comm, "_fused", False)
                                        else "p2p mailboxes, one exchange kernel per BatchNorm")
                if os.environ.get("CRIS_GRAD_EXCHANGE", "rccl") == "p2p":      # opt-in: the direct exchange over the mapped arenas
                    refused = comm.enable_arena_exchange(e_.grad_arena)
                    self.grad_exchange = "p2p" if refused is None else "rccl (arena exchange refused: %s)" % refused
            else:
                self._xgen_dev, self.syncbn_exchange = None, "collective (mailboxes refused: %s)" % why
        # parameter-layout gradient buffers for the tensors whose HIP gradient lives in the GEMM layout
        e = self._engine
        srcs, dsts, lays = [], [], []
        self._grad_out = {}
        self._grad_views_active = views
        for n, p in self._grad_params():
            lay = e.gemm_layout(n)
            if views:
                # gradients stay where the engine wrote them: a 3x3 / padded convolution weight's `.grad` is a strided view of
                # its GEMM-layout block [n][tap][cpad] (no conversion pass; in-place ops on `.grad` act on the arena)
                self._grad_out[n] = e.grad_param_view(n)
            elif lay is not None and not (lay[2] == 1 and lay[3] == lay[1]):
                d = torch.empty_like(p.data)
                srcs.append(e.G[n]); dsts.append(d); lays.append(lay)
                self._grad_out[n] = d
            else:
                self._grad_out[n] = e.G[n].view(p.shape)
        self._unpack = ops.UnpackTable(srcs, dsts, lays) if srcs else None
        return e

    def _direct_grads(self):
        """True when `.grad` may be assigned directly: no process group (DistributedDataParallel learns about gradients through
        autograd's AccumulateGrad hooks and needs them to flow through autograd), no hooks on any parameter, and not switched
        off (CRIS_MODULE_DIRECT_GRAD=0)."""
        if os.environ.get("CRIS_MODULE_DIRECT_GRAD", "1") != "1":
            return False
        if dist.is_available() and dist.is_initialized() and not getattr(self, "_self_exchange", False):
            # (a DDP wrapper that manages these parameters hooks their AccumulateGrad nodes, at any world size; with the
            # module's own exchange - _ddp_params_and_buffers_to_ignore - it has no hooks on them)
            return False
        if not torch.is_grad_enabled():
            return False
        return not any(p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None) for p in self._step_params)

    def _assign_grads(self, grads):
        """gradient semantics of autograd's accumulation on `.grad`: None -> the new gradient (the engine's own buffer, no copy);
        an existing tensor -> += (gradient accumulation over several backward passes).  `.grad` is never the engine's buffer
        here: _release_engine_grads() has replaced such a reference by a copy before this backward pass overwrote the buffer."""
        for p_, g in zip(self._step_params, grads):
            if not p_.requires_grad:
                continue                         # a frozen parameter keeps `.grad` None, as under autograd (optimizers skip it)
            if p_.grad is None:
                p_.grad = g
            else:
                p_.grad.add_(g)

    def _release_engine_grads(self):
        """Direct-gradient mode hands the engine's persistent buffers out as `.grad`.  A BACKWARD pass clears and re-uses them (a
        forward never touches the gradient arena), so whatever they still mean to the caller is saved first, at the start of
        backward: a parameter whose `.grad` is still the engine's buffer - zero_grad(set_to_none=False) leaves it in place
        (zeroed), and so does a loop that accumulates gradients over several micro-batches without any zero_grad - gets a copy
        instead; this backward then adds to the copy like autograd's AccumulateGrad would.  The reference's loop (forward,
        `optimizer.zero_grad()` - which sets `.grad` to None -, backward: engine/engine.py:48-55) never copies.  (Round 4, call
        r04f: doing this at the start of FORWARD copied all 449 gradients in every step of that loop - 2.5 ms.)"""
        bufs = self._step_grads
        for p_, g in zip(self._step_params, bufs):
            if p_.grad is g:
                p_.grad = g.clone()

    def _export_grads(self):
        if self._unpack is not None:
            self._unpack.run()
        return [self._grad_out[n] for n, _ in self._grad_params()]

    # ------------------------------------------------------------------------------------------------
    # HIP-graph replay of the training step under torch's autograd.  The engine's schedule is ~1000 launches; issued from
    # Python one by one it is host-bound (37.5 ms per R50 step against 13.8 for the native trainer, bench.py --path module,
    # round 3).  So the module captures, per input shape, TWO graphs over one memory pool - forward + loss (incl. the re-pack
    # of the bf16 operand copies: a torch optimizer changed the parameters) and backward + gradient export - and replays them
    # from _CrisStep.forward / .backward.  Step-varying scalars live in device memory: the dropout seed (Engine.seed_dev) and
    # GradScaler's factor (gscale).  First call of a shape runs eagerly (tables, allocator warm-up), the second captures.
    # Eager launches remain for: CRIS_MODULE_GRAPH=0, more than MAX_GRAPH_SHAPES shapes, a communicator whose collectives
    # cannot be captured (gloo), and any capture failure (reported once in `graph_error`).
    MAX_GRAPH_SHAPES = 6

    def _graph_step(self, img, word, mask, seed, xchg=None):
        """replay (capturing first if needed) the forward graph for this step; None = run the eager schedule"""
        if os.environ.get("CRIS_MODULE_GRAPH", "1") != "1":
            return None
        eng = self._engine
        if eng.comm.world > 1 and not getattr(eng.comm, "capturable", False):
            return None
        if getattr(self, "_graphs_key", None) != self._engine_key:           # parameters moved (.cuda() / .to()): start over
            self._graphs, self._graphs_key, self._graph_seen, self.graph_error = {}, self._engine_key, set(), None
        # (a backward with the gradient exchange inside and one without - the wrapper's no_sync() - are different schedules)
        key = (tuple(img.shape), tuple(word.shape), tuple(mask.shape), img.dtype, mask.dtype, word.dtype, xchg is not None)
        st = self._graphs.get(key)
        if st is None:
            if self.graph_error is not None or len(self._graphs) >= self.MAX_GRAPH_SHAPES:
                return None
            if key not in self._graph_seen:
                self._graph_seen.add(key)
                return None                                                  # first step with these shapes: eager
            if os.environ.get("CRIS_MODULE_REPLAY", "graph") == "cmdlist":
                # host command lists instead of HIP graphs: this call RUNS the forward while recording it
                st = self._record_forward(img, word, mask, seed)
                self._graphs[key] = st
                return st
            try:
                st = self._capture_graphs(img, word, mask, xchg)
            except Exception as ex:              # noqa: BLE001 - fall back to the eager schedule, say why once
                self.graph_error = repr(ex)
                torch.cuda.synchronize()
                eng.seed_dev = None
                return None
            self._graphs[key] = st
        if st.get("bwd") is None:
            raise RuntimeError("CRIS (HIP path): a training forward of a shape whose first backward has not run yet (command-list mode "
                               "records the backward pass with the first backward of a shape)")
        if not getattr(self, "_replay_repacks", True) and not eng.packs_current:
            eng.repack_weights()                                             # (weights changed outside the bound optimizer)
            eng.packs_current = True
        st["img"].copy_(img, non_blocking=True)
        st["word"].copy_(word, non_blocking=True)
        st["mask"].copy_(mask, non_blocking=True)
        st["seed"].fill_(((int(seed) & 0xFFFFFFFF) ^ 0x80000000) - 0x80000000)     # the uint32 seed's bit pattern in the int32 word
        st["fwd"].replay()
        return st

    def _prepare_packs_for_capture(self):
        """Whether a replayed forward re-packs the bf16 operand copies of the weights.  Under a torch optimizer it must (the
        optimizer changed the fp32 parameters behind the engine's back): the re-pack is part of every replay.  With
        cris.pytorch_amd.optim.Adam bound (gradient-view mode) the update itself rewrites the copies, so the replay carries no
        re-pack; a change from elsewhere (load_state_dict) clears `packs_current` and is repaired eagerly before the next replay."""
        eng = self._engine
        if self._grad_views_active:
            if not eng.packs_current:
                eng.repack_weights()
                eng.packs_current = True
            self._replay_repacks = False
        else:
            eng.packs_current = False
            self._replay_repacks = True

    def load_state_dict(self, *args, **kwargs):
        r = super().load_state_dict(*args, **kwargs)
        if self._engine is not None:
            self._engine.packs_current = False          # the bf16 operand copies no longer match the parameters
        return r

    def _record_forward(self, img, word, mask, seed):
        """CRIS_MODULE_REPLAY=cmdlist: the step's launches as host command lists (hip.CommandList) instead of two HIP graphs -
        every library call of one executed forward (here) and backward (_record_backward, from the first backward of the
        shape) is recorded with its arguments; all buffers come from a private memory pool that stays reserved.  A replay costs
        ~3 us of host time per launch but the first kernel starts at once - under a loop that synchronises every step
        (engine/engine.py:67-69) a graph launch's latency is paid twice per step and cannot be hidden by running ahead."""
        from .. import hip
        eng = self._engine
        st = dict(img=img.clone(), word=word.clone(), mask=mask.clone(),
                  seed=torch.zeros(1, dtype=torch.int32, device=img.device), gscale=torch.ones(1, dtype=torch.float32, device=img.device))
        st["seed"].fill_(((int(seed) & 0xFFFFFFFF) ^ 0x80000000) - 0x80000000)
        st["pool"] = torch.cuda.MemPool()
        fwd = hip.CommandList()
        eng.seed_dev = st["seed"]
        self._prepare_packs_for_capture()
        try:
            with torch.cuda.use_mem_pool(st["pool"]):
                hip.RECORDER = fwd
                try:
                    pred, msk, loss = self._training_forward(st["img"], st["word"], st["mask"], 0)
                finally:
                    hip.RECORDER = None
        finally:
            eng.seed_dev = None
        st.update(fwd=fwd, bwd=None, pred=pred, msk=msk, loss=loss)
        return st

    def _record_backward(self, st, xchg=None):
        from .. import hip
        eng = self._engine
        bwd = hip.CommandList()
        eng.seed_dev = st["seed"]
        try:
            with torch.cuda.use_mem_pool(st["pool"]):
                hip.RECORDER = bwd
                try:
                    self._backward_and_exchange(st["gscale"], xchg)
                    grads = self._export_grads()
                finally:
                    hip.RECORDER = None
        finally:
            eng.seed_dev = None
        st.update(bwd=bwd, grads=grads)

    def _capture_graphs(self, img, word, mask, xchg=None):
        eng = self._engine
        st = dict(img=img.clone(), word=word.clone(), mask=mask.clone(),
                  seed=torch.zeros(1, dtype=torch.int32, device=img.device), gscale=torch.ones(1, dtype=torch.float32, device=img.device))
        from .. import capture
        pool = torch.cuda.graph_pool_handle()
        fwd, bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        eng.seed_dev = st["seed"]
        try:
            self._prepare_packs_for_capture()
            # (capture.py: thread_local mode + drained c10d watchdog - DistributedDataParallel's all-reduces of the previous
            # step are still on the watchdog's list when the second step captures)
            with capture.graph(fwd, pool=pool):
                pred, msk, loss = self._training_forward(st["img"], st["word"], st["mask"], 0)
            with capture.graph(bwd, pool=pool, drain=False):
                self._backward_and_exchange(st["gscale"], xchg)
                grads = self._export_grads()
        finally:
            eng.seed_dev = None
        st.update(fwd=fwd, bwd=bwd, pred=pred, msk=msk, loss=loss, grads=grads)
        return st

    def forward(self, img, word, mask=None):
        """img: b, 3, h, w ; word: b, words ; mask: b, 1, h, w   (reference model/segmenter.py:29-35)"""
        if img.device.type != "cuda":
            raise RuntimeError("CRIS (HIP path) runs on the GPU only: got an input on %s - there is no CPU fallback" % img.device)
        eng = self._ensure_engine(img.device)
        if self.training:
            if mask is None:
                raise ValueError("training forward needs the mask")
            self._check_peers()
            seed = self._steps * 7919 + 17
            self._steps += 1
            if getattr(self, "_step_cache_key", None) != self._engine_key:       # (module traversals cost ~1 ms per step)
                self._step_params = [p for _, p in self._grad_params()]
                self._step_nbt = [m.num_batches_tracked for m in self.modules()
                                  if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.num_batches_tracked is not None]
                self._step_grads = [self._grad_out[n] for n, _ in self._grad_params()]
                self._anchor = torch.zeros(1, device=img.device, requires_grad=True)
                self._step_cache_key = self._engine_key
            if self._direct_grads():
                pred, msk, loss = _CrisStep.apply(self, img, word, mask, seed, self._anchor)
            else:
                pred, msk, loss = _CrisStep.apply(self, img, word, mask, seed, *self._step_params)
            # the running BatchNorm statistics were updated in place by the HIP kernels; keep torch's counters in step
            if self._step_nbt:
                torch._foreach_add_(self._step_nbt, 1)
            # (graph replay: pred / msk / loss are the capture's static buffers.  The loss gets a copy - a list of losses or a
            # deferred .item() must not change under the caller; pred and msk (0.35 MB each at 416x416, batch 8) stay aliases
            # that the NEXT training forward overwrites - the reference's loop consumes them before it, engine/engine.py:60-70)
            return pred.detach(), msk, loss.clone()
        with torch.no_grad():
            if os.environ.get("CRIS_EVAL_FOLD", "1") == "1":
                return self._eval_forward_folded(img, word)
            return eng.forward(img, word, None, training=False).detach()

    def invalidate_inference_cache(self):
        """Forget the folded weights / captured graph of the eval path.  The cache notices optimizer steps, load_state_dict and
        training forwards by itself (torch version counters, `_steps`); writes that bypass the version counter do NOT show
        (`p.data.copy_()`, `p.data.mul_()` in EMA or weight-surgery code, another engine updating the shared tensors through
        raw pointers) - call this after such a write, before the next `model.eval()` forward."""
        if getattr(self, "_infer", None) is not None:
            self._infer.invalidate()
        self._infer_sig = None

    def _eval_forward_folded(self, img, word):
        """`model.eval()` forward (engine/engine.py:100,171; test.py; tools/latency.py:62) on the inference engine: BatchNorms
        folded into their convolutions, one HIP graph per input shape (infer.py).  The folded weights are a cache of the
        parameters and running statistics: it is rebuilt when any of them changed since it was made - torch's version counters
        see optimizer steps and load_state_dict, `_steps` sees the running statistics the HIP training forward updates."""
        from ..infer import InferenceRunner
        eng = self._engine
        if getattr(self, "_ver_key", None) != self._engine_key:          # (the tensor objects change with .cuda() / .to())
            self._ver_tensors = list(self.parameters()) + list(self.buffers())
            self._ver_key = self._engine_key
        sig = (self._engine_key, self._steps, sum(t._version for t in self._ver_tensors))
        if getattr(self, "_infer", None) is None or self._infer_key != self._engine_key:
            self._infer = InferenceRunner(self.clip_spec, self.head_spec, None, img.device, tensors=(eng.P, eng.Bf))
            self._infer_key, self._infer_sig = self._engine_key, sig
        elif sig != self._infer_sig:
            self._infer.invalidate()
            self._infer_sig = sig
        return self._infer(img.float().contiguous(), word).clone()           # (the runner's buffer is overwritten by its next call)
