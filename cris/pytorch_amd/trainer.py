"""Native training step: Engine forward + backward, overlapped gradient all-reduce, fused HIP Adam.

Semantics follow the reference loop (engine/engine.py:37-73, train.py:105-111): Adam(betas 0.9/0.999, eps 1e-8,
weight_decay from the yaml) over two parameter groups built by the `build_segmenter` name rule
(model/__init__.py:36-48); bf16 needs no loss scaling so there is no GradScaler; the train metric
(utils/misc.py:114-129) is computed on the device without a host sync.
"""
from typing import Optional

import torch

from . import ops
from .arch import ClipSpec, HeadSpec
from .engine import Engine


def split_state_dict(sd, device):
    params = {k: v.to(device).contiguous() for k, v in sd.items()
              if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
    buffers = {k: v.to(device).contiguous() for k, v in sd.items() if k.endswith(("running_mean", "running_var"))}
    return params, buffers


class NativeTrainer:
    def __init__(self, clip: ClipSpec, head: HeadSpec, state_dict, device, base_lr=1e-4, lr_multi=0.1, weight_decay=0.0,
                 comm=None, sync_bn=False):
        self.device = device
        params, buffers = split_state_dict(state_dict, device)
        self.engine = Engine(clip, head, params, buffers, device, comm=comm, sync_bn=sync_bn)
        self.comm = self.engine.comm
        e = self.engine
        # `build_segmenter` groups: backbone (w/o positional embeddings) vs the rest.  torch's Adam is built with
        # lr=base_lr and groups that only carry `initial_lr`, so BOTH groups start at base_lr.
        names = [n for n in e.grad_order if n != "backbone.logit_scale"]          # never receives a gradient (unused)
        self.names = names
        self.group = {n: (0 if (n.startswith("backbone") and "positional_embedding" not in n) else 1) for n in names}
        self.base_lr, self.lr_multi, self.weight_decay = base_lr, lr_multi, weight_decay
        self.adam = ops.AdamTable([e.P[n] for n in names], [e.G[n] for n in names], [base_lr] * len(names),
                                  layouts=[e.gemm_layout(n) for n in names])
        self.step_idx = 0
        self.metric = torch.zeros(2, device=device)

    def set_group_lrs(self, lr_backbone, lr_head):
        self.adam.set_lrs([lr_backbone if self.group[n] == 0 else lr_head for n in self.names])

    def train_step(self, img, word, mask, seed: Optional[int] = None):
        e = self.engine
        seed = self.step_idx * 7919 + 17 if seed is None else seed
        pred, msk, loss = e.forward(img, word, mask, training=True, seed=seed)
        if self.comm.world > 1:
            def on_stage(st):
                lo, hi = e.stage_ranges[st]
                self.comm.allreduce_async(e.grad_arena[lo:hi])
            e.backward(on_stage_done=on_stage)
            self.comm.wait_all()
        else:
            e.backward()
        self.adam.step(weight_decay=self.weight_decay, grad_scale=1.0 / self.comm.world)
        self.metric.zero_()
        ops.train_metric(pred, msk, pred.shape[0], pred.shape[2] * pred.shape[3], self.metric)
        self.step_idx += 1
        return loss, self.metric

    @torch.no_grad()
    def eval_forward(self, img, word):
        return self.engine.forward(img, word, None, training=False)
