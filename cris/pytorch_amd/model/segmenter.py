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

import torch
import torch.distributed as dist
from torch import nn

from .. import arch, ops
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
        pred, msk, loss = eng.forward(img, word, mask, training=True, seed=seed)
        ctx.module = module
        ctx.mark_non_differentiable(pred, msk)
        return pred, msk, loss

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, _gpred, _gmsk, gloss):
        module = ctx.module
        eng = module._engine
        gscale = gloss.detach().reshape(1).to(torch.float32).contiguous()        # GradScaler's factor arrives here
        eng.backward(gscale=gscale)
        grads = module._export_grads()
        return (None, None, None, None, None) + tuple(grads)


class CRIS(nn.Module):
    def __init__(self, cfg):
        super().__init__()
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
        self._steps = 0
        self._unpack = None

    # ------------------------------------------------------------------------------------------------
    def _grad_params(self):
        return [(n, p) for n, p in self.named_parameters() if n != "backbone.logit_scale"]

    def _ensure_engine(self, device):
        params = {n: p.data for n, p in self.named_parameters()}
        buffers = {n: b for n, b in self.named_buffers() if n.endswith(("running_mean", "running_var"))}
        sync = (any(isinstance(m, nn.SyncBatchNorm) for m in self.modules()) and dist.is_available()
                and dist.is_initialized() and dist.get_world_size() > 1)
        key = (str(device), sync, tuple(p.data_ptr() for p in params.values()), tuple(b.data_ptr() for b in buffers.values()))
        if self._engine is not None and key == self._engine_key:
            return self._engine
        for n, p in params.items():
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("CRIS (HIP path) needs contiguous fp32 parameters, got %s %s" % (n, p.dtype))
        comm = None
        if sync:
            from ..dist import TorchDistComm
            comm = TorchDistComm(device)
        self._engine = Engine(self.clip_spec, self.head_spec, params, buffers, device, comm=comm, sync_bn=sync)
        self._engine_key = key
        # parameter-layout gradient buffers for the tensors whose HIP gradient lives in the GEMM layout
        e = self._engine
        srcs, dsts, lays = [], [], []
        self._grad_out = {}
        for n, p in self._grad_params():
            lay = e.gemm_layout(n)
            if lay is not None and not (lay[2] == 1 and lay[3] == lay[1]):
                d = torch.empty_like(p.data)
                srcs.append(e.G[n]); dsts.append(d); lays.append(lay)
                self._grad_out[n] = d
            else:
                self._grad_out[n] = e.G[n].view(p.shape)
        self._unpack = ops.UnpackTable(srcs, dsts, lays) if srcs else None
        return e

    def _export_grads(self):
        if self._unpack is not None:
            self._unpack.run()
        return [self._grad_out[n] for n, _ in self._grad_params()]

    def forward(self, img, word, mask=None):
        """img: b, 3, h, w ; word: b, words ; mask: b, 1, h, w   (reference model/segmenter.py:29-35)"""
        if img.device.type != "cuda":
            raise RuntimeError("CRIS (HIP path) runs on the GPU only: got an input on %s - there is no CPU fallback" % img.device)
        eng = self._ensure_engine(img.device)
        if self.training:
            if mask is None:
                raise ValueError("training forward needs the mask")
            seed = self._steps * 7919 + 17
            self._steps += 1
            params = [p for _, p in self._grad_params()]
            pred, msk, loss = _CrisStep.apply(self, img, word, mask, seed, *params)
            # the running BatchNorm statistics were updated in place by the HIP kernels; keep torch's counters in step
            nbt = [m.num_batches_tracked for m in self.modules()
                   if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.num_batches_tracked is not None]
            if nbt:
                torch._foreach_add_(nbt, 1)
            return pred.detach(), msk, loss
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
