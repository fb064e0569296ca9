"""TEST INFRASTRUCTURE ONLY - the oracle as a trainer: forward (oracle/cris_oracle.py) + autograd + torch.optim.Adam on any
device, in one of four precision policies:

  "fp64"   every parameter, buffer and activation in float64 (the oracle is dtype-generic).  The teacher of the teacher-forced
           parity tests since round 5: at the 1e-3 scale of the north star its trajectory does not depend on which MIOpen /
           hipBLASLt kernel a box picks or on the summation order of atomics (an fp32 teacher's trajectory through the untrained
           head's first 40 steps moved the 100-state mean |dloss| by +-25 % from box to box, profiles/parity_r04.md);
  "fp32"   plain fp32 (on the CPU this is the pinned ground truth; on the GPU it is checked equal to it,
           tests/test_oracle_device.py);
  "fp16"   the REFERENCE's own policy: torch.autocast(float16) around forward + loss, GradScaler around backward and the
           optimizer step (engine/engine.py:48-57, `amp.GradScaler()` engine/engine.py:27);
  "bf16"   torch.autocast(bfloat16), no scaler.

Under autocast BatchNorm / LayerNorm go through F.batch_norm / F.layer_norm (cris_oracle.NATIVE_NORMS) - the operators the
reference's modules call - so every operator follows autocast's own cast policy: this is "stock PyTorch on this hardware",
the yardstick the HIP path's distance to the fp32 oracle is read against (tools/parity_study.py), and the teacher of the
teacher-forced trajectory test (tests/test_engine_gpu.py).  Dropout masks come from the shared counter hash in every mode.
Adam: lr for both parameter groups = base lr (what the reference's first epoch does)."""
import contextlib
import dataclasses

import torch

from . import cris_oracle as O


def seed_of_step(t):
    """dropout seed of optimizer step t - the rule of tests/golden/make_trajectory.py and of the native trainer's device counter"""
    return t * 7919 + 17


class OracleTrainer:
    def __init__(self, clip, head, sd, device, mode="fp32", lr=1e-4):
        assert mode in ("fp64", "fp32", "fp16", "bf16")
        self.clip, self.head, self.device, self.mode = clip, head, torch.device(device), mode
        self.leaf = {}
        for k, v in sd.items():
            t = v.detach().to(self.device).clone()
            if mode == "fp64" and t.is_floating_point():
                t = t.double()
            if t.is_floating_point() and not k.endswith(("running_mean", "running_var")):
                t.requires_grad_(True)
            self.leaf[k] = t
        self.params = {k: v for k, v in self.leaf.items() if v.requires_grad}
        self.opt = torch.optim.Adam(list(self.params.values()), lr=lr)
        self.scaler = torch.amp.GradScaler(self.device.type) if mode == "fp16" else None
        if self.device.type == "cuda":
            torch.backends.cudnn.allow_tf32 = False
            torch.backends.cuda.matmul.allow_tf32 = False
            # the teacher must be reproducible: at the steep states of the untrained head (steps 3, 8, 10 of configs[1]) a
            # run-to-run difference of its atomics-based backward kernels moved the 100-state mean |dloss| by several per cent
            torch.backends.cudnn.deterministic = True
            torch.backends.cudnn.benchmark = False
            torch.use_deterministic_algorithms(True, warn_only=True)

    def _ctx(self):
        if self.mode in ("fp32", "fp64"):
            return contextlib.nullcontext()
        return torch.autocast(self.device.type, dtype=torch.float16 if self.mode == "fp16" else torch.bfloat16)

    def forward_backward(self, batch, seed):
        """loss (python float), logits, {name: gradient} of one batch at the current state; no update"""
        img, word, mask = (t.to(self.device) for t in batch)
        self.opt.zero_grad(set_to_none=True)
        bnu = {}
        O.NATIVE_NORMS = self.mode not in ("fp32", "fp64")
        try:
            with self._ctx():
                pred, m, loss = O.cris_forward(self.leaf, self.clip, self.head, img, word, mask, training=True,
                                               drop_seed=seed if self.head.dropout > 0 else None, bn_updates=bnu)
            if self.scaler is not None:
                self.scaler.scale(loss).backward()
            else:
                loss.backward()
        finally:
            O.NATIVE_NORMS = False
        self._bnu, self._pred, self._mask = bnu, pred.detach(), m
        return float(loss.detach()), pred.detach().float()

    def grads(self):
        """gradients of the last forward_backward, unscaled copies (the optimizer's own buffers stay scaled for scaler.step)"""
        s = 1.0 if self.scaler is None else 1.0 / float(self.scaler.get_scale())
        return {k: (v.grad.detach().float() * s) for k, v in self.params.items() if v.grad is not None}

    def update(self):
        if self.scaler is not None:
            self.scaler.step(self.opt)
            self.scaler.update()
        else:
            self.opt.step()
        with torch.no_grad():
            for pfx, (rm, rv) in self._bnu.items():
                self.leaf[pfx + ".running_mean"].copy_(rm)
                self.leaf[pfx + ".running_var"].copy_(rv)

    def step(self, batch, seed):
        loss, _ = self.forward_backward(batch, seed)
        self.update()
        return loss

    def metric(self):
        return O.train_metric(self._pred.float(), self._mask)

    def state_dict(self):
        return {k: v.detach() for k, v in self.leaf.items()}


def cosines(ga, gb, skip=("k_proj.bias",)):
    """{name: cosine} over the tensors both dicts hold (k-projection biases have an analytically zero gradient: skipped)"""
    out = {}
    for k, a in ga.items():
        b = gb.get(k)
        if b is None or k.endswith(skip):
            continue
        a, b = a.double().flatten(), b.double().flatten().to(a.device)
        na, nb = a.norm(), b.norm()
        if float(nb) == 0.0 or float(na) == 0.0:
            continue
        out[k] = float((a @ b) / (na * nb))
    return out


def with_dropout(head, p):
    return dataclasses.replace(head, dropout=p)
