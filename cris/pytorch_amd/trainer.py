"""Native training step: Engine forward + backward, overlapped gradient all-reduce, fused HIP Adam.

Semantics follow the reference loop (engine/engine.py:37-73, train.py:105-111): Adam(betas 0.9/0.999, eps 1e-8,
weight_decay from the yaml) over two parameter groups built by the `build_segmenter` name rule
(model/__init__.py:36-48); bf16 needs no loss scaling so there is no GradScaler; the train metric
(utils/misc.py:114-129) is computed on the device without a host sync.

Launch model: the step is a static schedule of ~1000 kernel launches.  After one eager step (which fills the host-side
caches and the allocator) the whole step - forward, backward, gradient exchange, Adam, metric - is captured ONCE into a
HIP graph (torch.cuda.CUDAGraph over the launch stream; the independent text-encoder branch is captured on a second
stream and so becomes a parallel branch of the graph) and replayed per step: one host call per step instead of one per
kernel.  What changes from step to step lives in device memory: the input batch (static buffers the caller's tensors
are copied into), the step counter (Adam bias corrections) and the dropout seed (`cris_step_advance`).

With more than one rank the step contains RCCL collectives (SyncBN statistics, gradient exchange).  There the default
launch mode is a host-side COMMAND LIST (hip.CommandList): the second step is executed once more by the Python schedule
while every library call (function pointer + ctypes arguments) and every torch-level op (stream wait, collective) is
recorded, with all of its buffers allocated from a private torch MemPool so their addresses stay valid; later steps
replay the list - a few microseconds of Python per launch, collectives issued by torch.distributed as usual.
`launch=` / CRIS_LAUNCH selects "graph", "cmdlist" or "eager" explicitly.
"""
import os
from typing import Optional

import torch

from . import debug, ops
from .arch import ClipSpec, HeadSpec
from .engine import Engine


def strip_ddp_prefix(sd):
    """The reference saves `model.state_dict()` of the DistributedDataParallel-wrapped model (train.py:192-204), so every key
    of its checkpoints starts with `module.`, and test.py:74-78 loads them strictly into a DataParallel wrapper.  Accept
    both spellings: a state_dict whose keys ALL carry the prefix is returned without it, anything else unchanged."""
    keys = list(sd.keys())
    if keys and all(k.startswith("module.") for k in keys):
        return {k[len("module."):]: v for k, v in sd.items()}
    return sd


def split_state_dict(sd, device):
    sd = strip_ddp_prefix(sd)
    params = {k: v.to(device).contiguous() for k, v in sd.items()
              if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
    buffers = {k: v.to(device).contiguous() for k, v in sd.items() if k.endswith(("running_mean", "running_var"))}
    return params, buffers


def epoch_group_lrs(epoch, base_lr, lr_multi, milestones, gamma):
    """(lr_backbone, lr_head) the reference trains epoch `epoch` (0-based) with - train.py:105-110,210.
    Adam is built with lr=base_lr over groups that carry only `initial_lr`, so epoch 0 runs BOTH groups at base_lr;
    `scheduler.step(epoch_log)` with an explicit epoch then switches MultiStepLR to its closed form on `initial_lr`
    (= lr_multi*base_lr for the backbone group): lr_e = initial_lr * gamma ** #{m in milestones : m <= e} for e >= 1."""
    if epoch <= 0:
        return base_lr, base_lr
    f = gamma ** sum(1 for m in milestones if m <= epoch)
    return lr_multi * base_lr * f, base_lr * f


class NativeTrainer:
    def __init__(self, clip: ClipSpec, head: HeadSpec, state_dict, device, base_lr=1e-4, lr_multi=0.1, weight_decay=0.0,
                 comm=None, sync_bn=False, use_graph: Optional[bool] = None, launch: Optional[str] = None):
        self.device = device
        params, buffers = split_state_dict(state_dict, device)
        self.engine = Engine(clip, head, params, buffers, device, comm=comm, sync_bn=sync_bn)
        self.comm = self.engine.comm
        e = self.engine
        if self.comm.world > 1:
            # DistributedDataParallel broadcasts rank 0's parameters and buffers when it wraps the module (train.py:100-102);
            # the single-exchange SyncBN takes its moments about the running mean, which must be bit-identical on every rank
            for t in list(e.P.values()) + list(e.Bf.values()):
                self.comm.broadcast(t, 0)
        self._checked_batch = None
        # `build_segmenter` groups: backbone (w/o positional embeddings) vs the rest.  torch's Adam is built with
        # lr=base_lr and groups that only carry `initial_lr`, so BOTH groups start at base_lr.
        names = [n for n in e.grad_order if n != "backbone.logit_scale"]          # never receives a gradient (unused)
        self.names = names
        self.group = {n: (0 if (n.startswith("backbone") and "positional_embedding" not in n) else 1) for n in names}
        self.base_lr, self.lr_multi, self.weight_decay = base_lr, lr_multi, weight_decay
        # Rows of the token embedding (49408 x 512: 17% of the parameters) that have never received a gradient keep g = m = v = 0
        # and Adam leaves them exactly as they are (while weight_decay == 0): the embedding backward marks the rows of each
        # batch's tokens and the update skips the rest - bit-identical to the dense update.  With more ranks the all-reduced
        # gradient has the other ranks' rows too: the marks (sticky bytes) are then all-reduced with MAX next to the text
        # encoder's gradient stage (round 6; 49 KB per step on the gradient communicator), so every rank skips exactly the rows
        # no rank ever touched - communicators without a byte-wise MAX (dist.RcclComm) keep the dense update.
        # CRIS_ADAM_ROW_SKIP=0 switches it off.
        if ((self.comm.world == 1 or getattr(self.comm, "supports_max_u8", False)) and weight_decay == 0.0
                and os.environ.get("CRIS_ADAM_ROW_SKIP", "1") == "1" and torch.device(device).type == "cuda"):
            e.embed_live = torch.zeros(e.P["backbone.token_embedding.weight"].shape[0], dtype=torch.uint8, device=device)
        self._build_adam([base_lr] * len(names))
        self.metric = torch.zeros(2, device=device)
        # per-step device state: steps done (int32) and the dropout seed of the running step
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=device)
        self.seed_dev = torch.zeros(1, dtype=torch.int32, device=device)
        # generation of the peer-mailbox exchanges: advances with every step and is NEVER rewritten (the optimizer step is, by
        # load_optimizer_state_dict: a rewound generation would accept the stale words of its first use, csrc/p2p_ll.h)
        self.xgen_dev = torch.zeros(1, dtype=torch.int32, device=device)
        # SyncBN statistics: 142 exchanges of a few KB per step, all on the critical path.  Default: peer-mapped mailboxes, one
        # kernel per exchange (dist.PeerMailboxes) - when allocation, IPC mapping and a self-test with known data succeed on
        # EVERY rank; otherwise (and with CRIS_SYNCBN_P2P=0, or a communicator without mailboxes) the RCCL collectives.
        self.syncbn_exchange = "none" if not e.sync_bn else "collective"
        if (e.sync_bn and os.environ.get("CRIS_SYNCBN_P2P", "1") == "1"          # (world 1 + CRIS_FORCE_DIST: the exchange with itself)
                and torch.device(device).type == "cuda" and hasattr(self.comm, "enable_p2p")):
            cmax = max(e.P[pfx + ".weight"].numel() for pfx in e.bn_prefixes)
            why = self.comm.enable_p2p(slots=2 * len(e.bn_prefixes) + 8, max_floats=4 * cmax, gen_dev=self.xgen_dev)
            if why is None:
                fused = getattr(self.comm, "_fused", False)
                self.syncbn_exchange = "p2p mailboxes, exchanged inside the BatchNorm launches" if fused else "p2p mailboxes, one exchange kernel per BatchNorm"
            else:
                self.syncbn_exchange = "collective (mailboxes refused: %s)" % why
        # gradient exchange: RCCL through torch.distributed (the default: eight staged all-reduces on their own communicator and
        # stream); CRIS_GRAD_EXCHANGE=p2p: the direct reduce-scatter + all-gather over the peer-mapped gradient arenas
        # (dist.TorchDistComm.enable_arena_exchange) when mapping and self-test succeed on every rank, else RCCL
        self.grad_exchange = "none" if (self.comm.world == 1 and not debug.HOOKS.force_dist) else "rccl"
        if (os.environ.get("CRIS_GRAD_EXCHANGE", "rccl") == "p2p" and getattr(self.comm, "p2p", None) is not None
                and hasattr(self.comm, "enable_arena_exchange")):
            why = self.comm.enable_arena_exchange(e.grad_arena)
            self.grad_exchange = ("p2p: reduce-scatter + all-gather over the peer-mapped gradient arenas" if why is None
                                  else "rccl (arena exchange refused: %s)" % why)
        if launch is None:
            launch = os.environ.get("CRIS_LAUNCH")
        if launch is None:
            if use_graph is False or os.environ.get("CRIS_NO_GRAPH", "0") == "1":
                launch = "eager"
            else:
                # one captured HIP graph per step on any number of ranks: RCCL's kernels are captured like every other
                # launch (measured with a 1-rank RCCL group, tools/dist1_check.py: 15.4 ms/step captured against 16.9 as a
                # command list, ~145 collectives per step); a capture that fails on ANY rank makes every rank fall back
                # to the command list (train_step).  Communicators that cannot be captured (gloo: host copies) say so.
                launch = "graph" if getattr(self.comm, "capturable", True) else "cmdlist"
        if torch.device(device).type != "cuda":
            launch = "eager"
        assert launch in ("graph", "cmdlist", "eager"), launch
        self.launch = launch
        self.use_graph = launch != "eager"
        self._graph = None
        self._cmds = None
        self._pool = None
        self._static = None
        self._eager_steps = 0
        self.graph_error = None
        self._host_steps = 0
        self._peer_check_every = int(os.environ.get("CRIS_PEER_CHECK_EVERY", "200"))

    def _build_adam(self, lrs):
        e, names = self.engine, self.names
        lr_of = dict(zip(names, lrs))
        live = {names.index("backbone.token_embedding.weight"): e.embed_live} if e.embed_live is not None else None
        self.adam = ops.AdamTable([e.P[n] for n in names], [e.G[n] for n in names], [lr_of[n] for n in names],
                                  layouts=[e.gemm_layout(n) for n in names], packs=[e.pack_info.get(n) for n in names], row_live=live)

    def check_peer_timeout(self):
        """collective: raise on every rank if any rank's SyncBN mailbox exchange timed out (dist.TorchDistComm.check_peer_timeout)"""
        chk = getattr(self.comm, "check_peer_timeout", None)
        if chk is not None:
            chk()

    @property
    def step_idx(self):
        return int(self.step_dev.item())

    def set_group_lrs(self, lr_backbone, lr_head):
        lrs = [lr_backbone if self.group[n] == 0 else lr_head for n in self.names]
        self.adam.set_lrs(lrs)
        self._graph = self._cmds = None          # learning rates live in the device table, which was re-uploaded (new
        self._eager_steps = 0                    # address): capture / record again

    def set_epoch(self, epoch, milestones=(35,), gamma=0.1):
        """Learning rates of the reference schedule for `epoch` (0-based); call at every epoch boundary."""
        self.set_group_lrs(*epoch_group_lrs(epoch, self.base_lr, self.lr_multi, milestones, gamma))

    # ------------------------------------------------------------------------------------------------
    def _step_body(self, img, word, mask, host_seed: Optional[int]):
        e = self.engine
        if host_seed is None:
            ops.step_advance(self.step_dev, self.seed_dev, self.xgen_dev)
            e.seed_dev, seed = self.seed_dev, 0
        else:                                    # explicit seed (tests): host value, the device counter still advances
            ops.step_advance(self.step_dev, self.seed_dev, self.xgen_dev)
            e.seed_dev, seed = None, host_seed
        pred, msk, loss = e.forward(img, word, mask, training=True, seed=seed)
        # the train metric (utils/misc.py:114-129) only needs the logits: it runs on the text-encoder stream underneath the
        # backward pass (backward() joins that stream before it returns)
        if e.side is not None:
            cur = torch.cuda.current_stream()
            ops.torch_op(lambda: e.side.wait_stream(cur))
            with torch.cuda.stream(e.side):
                ops.train_metric(pred, msk, pred.shape[0], pred.shape[2] * pred.shape[3], self.metric)
        else:
            ops.train_metric(pred, msk, pred.shape[0], pred.shape[2] * pred.shape[3], self.metric)
        if self.comm.world > 1 or debug.HOOKS.force_dist:
            def on_stage(st):
                lo, hi = e.stage_ranges[st]
                ops.torch_op(lambda: self.comm.allreduce_async(e.grad_arena[lo:hi]))
                if st == 4 and e.embed_live is not None and getattr(self.comm, "supports_max_u8", False):
                    # (stage 4 = the text encoder, whose backward marked this batch's rows of the token embedding)
                    ops.torch_op(lambda: self.comm.allreduce_async(e.embed_live, op="max"))
            e.backward(on_stage_done=on_stage)
            ops.torch_op(self.comm.wait_all)
        else:
            e.backward()
        # one Adam pass over every tensor; it also rewrites the bf16 operand copies of the GEMM weights from the new values
        self.adam.step(weight_decay=self.weight_decay, grad_scale=1.0 / self.comm.world, step_dev=self.step_dev)
        e.packs_current = self.adam.refreshes_packs
        return loss, pred, msk

    def train_step(self, img, word, mask, seed: Optional[int] = None):
        """One optimizer step.  Returns (loss 0-dim device tensor, metric [IoU%, Pr@50%] device tensor); both are
        overwritten by the next call."""
        # COLLECTIVE, every CRIS_PEER_CHECK_EVERY-th step (default 200; 0 = never): a rank whose SyncBN mailbox exchange gave up
        # waiting for a peer raises on EVERY rank instead of training on alone (round-5 advisor finding: nothing called the check)
        self._host_steps += 1
        if self._peer_check_every > 0 and self._host_steps % self._peer_check_every == 0 and getattr(self.comm, "p2p", None) is not None:
            self.check_peer_timeout()
        if not self.use_graph or seed is not None or ops.KERNEL_TIMER is not None:
            self._check_equal_batch(img.shape[0])
            loss, _, _ = self._step_body(img, word, mask, seed)
            return loss, self.metric
        key = (tuple(img.shape), tuple(word.shape), tuple(mask.shape))
        self._check_equal_batch(img.shape[0])
        if self._static is not None and self._static[0] != key:
            self._graph, self._cmds, self._static, self._eager_steps = None, None, None, 0      # new shapes: new schedule
        if self._graph is None and self._cmds is None:
            if self._eager_steps < 1:
                # first step with these shapes runs eagerly: constant tables get uploaded, the allocator warms up
                self._eager_steps += 1
                self._static = (key, img.clone(), word.clone(), mask.clone())
                loss, _, _ = self._step_body(self._static[1], self._static[2], self._static[3], None)
                return loss, self.metric
            if self.launch == "cmdlist":
                _, s_img, s_word, s_mask = self._static
                s_img.copy_(img, non_blocking=True)
                s_word.copy_(word, non_blocking=True)
                s_mask.copy_(mask, non_blocking=True)
                return self._record(), self.metric            # this call executes the step while recording it
            err = None
            try:
                self._capture()
            except Exception as ex:              # noqa: BLE001 - e.g. a collective that cannot be captured
                err = repr(ex)
            if self.comm.world > 1:              # the ranks must agree on the launch mode
                err = next((x for x in self.comm.all_gather_object(err) if x), None)
            if err is not None:
                self.graph_error, self._graph = err, None
                torch.cuda.synchronize(self.device)
                if self.comm.world > 1:
                    self.launch = "cmdlist"
                    _, s_img, s_word, s_mask = self._static
                    s_img.copy_(img, non_blocking=True)
                    s_word.copy_(word, non_blocking=True)
                    s_mask.copy_(mask, non_blocking=True)
                    return self._record(), self.metric
                self.use_graph = False
                self.launch = "eager"
                loss, _, _ = self._step_body(img, word, mask, None)
                return loss, self.metric
        _, s_img, s_word, s_mask = self._static
        s_img.copy_(img, non_blocking=True)
        s_word.copy_(word, non_blocking=True)
        s_mask.copy_(mask, non_blocking=True)
        if self._graph is not None:
            self._graph.replay()
        else:
            self._cmds.replay()
        return self._loss, self.metric

    def _check_equal_batch(self, b):
        """SyncBN's global count is local count x world: every rank must feed the same per-rank batch (the reference's
        DistributedSampler + fixed DataLoader batch size guarantee it; checked once per batch size, not per step)."""
        if self.comm.world > 1 and self._checked_batch != b:
            sizes = self.comm.all_gather_object(int(b))
            if any(x != sizes[0] for x in sizes):
                raise ValueError("per-rank batch sizes differ across ranks (%s): SyncBN statistics assume equal shards" % (sizes,))
            self._checked_batch = b

    def _capture(self):
        from . import capture
        _, s_img, s_word, s_mask = self._static
        g = torch.cuda.CUDAGraph()
        # thread_local capture mode + a drained c10d watchdog (capture.py): the eager first step left collectives of two
        # communicators behind whose end events the watchdog thread is still polling
        with capture.graph(g, device=self.device):
            loss, pred, msk = self._step_body(s_img, s_word, s_mask, None)
        self._graph, self._loss, self._keep = g, loss, (pred, msk)

    def _record(self):
        """Execute one step through the Python schedule while recording it as a command list; every buffer it allocates
        comes from a private MemPool that stays reserved, so the recorded addresses remain valid for the replays."""
        from . import hip
        _, s_img, s_word, s_mask = self._static
        self._pool = torch.cuda.MemPool()
        rec = hip.CommandList()
        with torch.cuda.use_mem_pool(self._pool):
            hip.RECORDER = rec
            try:
                loss, pred, msk = self._step_body(s_img, s_word, s_mask, None)
            finally:
                hip.RECORDER = None
        self._cmds, self._loss, self._keep = rec, loss, (pred, msk)
        return loss

    # ------------------------------------------------------------------------------------------------
    # checkpointing (reference train.py:159-174,192-207: {'epoch', 'cur_iou', 'best_iou', 'state_dict', 'optimizer', 'scheduler'})
    # ------------------------------------------------------------------------------------------------
    def _param_order(self):
        """parameter names in the order torch.optim.Adam(param_list) numbers them for `build_segmenter`'s two groups
        (model/__init__.py:36-48): group 0 = backbone without the positional embeddings, group 1 = the rest - module order
        inside each group.  `backbone.logit_scale` is a parameter of the reference module too (it just never gets a gradient)."""
        names = list(self.engine.P.keys())                    # state_dict (= named_parameters) order
        g0 = [n for n in names if n.startswith("backbone") and "positional_embedding" not in n]
        g1 = [n for n in names if n not in set(g0)]
        return g0, g1

    def model_state_dict(self, ddp_prefix=False):
        """the reference module's `state_dict()` (parameters + BatchNorm buffers; clones, on the CPU).  `ddp_prefix=True`:
        keys spelled `module.<name>` like the checkpoints the reference writes from its DDP-wrapped model
        (train.py:192-204) - what its `--resume` (train.py:159-174) and test.py:74-78 load strictly."""
        e = self.engine
        out = {k: v.detach().cpu().clone() for k, v in e.P.items()}
        out.update({k: v.detach().cpu().clone() for k, v in e.Bf.items()})
        steps = self.step_idx
        for pfx in e.bn_prefixes:
            out[pfx + ".num_batches_tracked"] = torch.tensor(steps, dtype=torch.int64)
        # reference key order (module order), not "parameters then buffers"
        from .arch import build_param_tree
        order = list(build_param_tree(e.clip, e.head).state_dict().keys())
        assert set(order) == set(out.keys())
        out = {k: out[k] for k in order}
        if ddp_prefix:
            out = {"module." + k: v for k, v in out.items()}
        return out

    def load_model_state_dict(self, sd):
        """parameters and BatchNorm buffers from a reference-keyed state_dict; the bf16 operand copies are re-packed on the
        next forward.  Keys may carry DDP's `module.` prefix (reference checkpoints do)."""
        e = self.engine
        sd = strip_ddp_prefix(sd)
        missing = [k for k in list(e.P) + list(e.Bf) if k not in sd]
        if missing:
            raise KeyError("state_dict lacks %d keys, e.g. %s" % (len(missing), missing[:3]))
        for k, t in list(e.P.items()) + list(e.Bf.items()):
            t.copy_(sd[k].to(self.device))
        e.packs_current = False

    def optimizer_state_dict(self):
        """Adam state in torch.optim.Adam.state_dict() form, loadable by the optimizer `train.py:105-107` builds from
        `build_segmenter`'s param_list (and by load_optimizer_state_dict)."""
        g0, g1 = self._param_order()
        idx = {n: i for i, n in enumerate(self.names)}
        lr_of = dict(zip(self.names, self.adam.lrs))
        state, step = {}, self.step_idx
        for i, n in enumerate(g0 + g1):
            if n in idx and step > 0:
                j = idx[n]
                state[i] = {"step": torch.tensor(float(step)), "exp_avg": self.adam.m[j].detach().cpu().clone(),
                            "exp_avg_sq": self.adam.v[j].detach().cpu().clone()}
        def group(names, initial_lr, first):
            lr = next((lr_of[n] for n in names if n in lr_of), self.base_lr)
            return {"lr": lr, "initial_lr": initial_lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": self.weight_decay,
                    "amsgrad": False, "params": list(range(first, first + len(names)))}
        return {"state": state, "param_groups": [group(g0, self.lr_multi * self.base_lr, 0), group(g1, self.base_lr, len(g0))]}

    def load_optimizer_state_dict(self, sd):
        """restore m / v / step count (and the group learning rates) from optimizer_state_dict() or from the state_dict of a
        torch.optim.Adam over the same param_list"""
        g0, g1 = self._param_order()
        idx = {n: i for i, n in enumerate(self.names)}
        step = 0
        for i, n in enumerate(g0 + g1):
            st = sd["state"].get(i)
            if st is None or n not in idx:
                continue
            j = idx[n]
            self.adam.m[j].copy_(st["exp_avg"].to(self.device))
            self.adam.v[j].copy_(st["exp_avg_sq"].to(self.device))
            step = max(step, int(float(st["step"])))
        for j, live in self.adam.row_live.items():       # rows with Adam state are live rows
            live.copy_(((self.adam.m[j] != 0) | (self.adam.v[j] != 0)).any(dim=1))
        self.step_dev.fill_(step)
        self.adam.step_count = step
        self.set_group_lrs(sd["param_groups"][0]["lr"], sd["param_groups"][1]["lr"])

    @torch.no_grad()
    def eval_forward(self, img, word):
        return self.engine.forward(img, word, None, training=False)
