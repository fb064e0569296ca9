"""GPU tests of the drop-in nn.Module (cris.pytorch_amd.model.CRIS) driven the way the reference's engine drives its model
(engine/engine.py:37-57): ambient fp16 autocast, GradScaler, torch.optim.Adam over build_segmenter's groups."""
from types import SimpleNamespace as NS

import pytest
import torch

pytestmark = pytest.mark.gpu

from cris.pytorch_amd import arch, synth  # noqa: E402
from cris.pytorch_amd.model import build_segmenter  # noqa: E402
from cris.pytorch_amd.trainer import NativeTrainer  # noqa: E402
from test_module_surface import TINY  # noqa: E402


def _batch(step, dev):
    return tuple(t.to(dev) for t in synth.make_batch(4, 64, 9, 0, step))


def test_module_train_loop_matches_native_trainer():
    dev = torch.device("cuda:0")
    model, groups = build_segmenter(NS(**TINY))
    clip, head = arch.specs_by_name("tiny")
    # the head's torch-default init has BatchNorm beta = 0, which makes several BN gammas / betas exactly scale- or
    # shift-invariant (analytically zero gradients); Adam turns the rounding noise of such gradients into +-lr steps and two
    # runs of anything drift apart.  The trained-like synthetic state keeps the comparison meaningful.
    model.load_state_dict(arch.synthetic_state_dict(clip, head, 0))
    model = model.to(dev).train()
    opt = torch.optim.Adam(groups, lr=1e-4, weight_decay=0.0)       # as train.py:105-107: both groups start at base_lr
    scaler = torch.amp.GradScaler("cuda")
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    tr = NativeTrainer(clip, head, sd, dev, base_lr=1e-4, use_graph=False)
    for step in range(4):
        img, word, mask = _batch(step, dev)
        with torch.autocast("cuda"):                                # engine/engine.py:48 (fp16 autocast is ambient)
            pred, target, loss = model(img, word, mask)
        opt.zero_grad()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        ref_loss, _ = tr.train_step(img, word, mask)
        assert pred.shape == (4, 1, 16, 16) and target.shape == pred.shape and loss.dim() == 0 and loss.requires_grad
        assert not pred.requires_grad
        # same kernels, same seeds, same Adam arithmetic (torch's vs the fused HIP one): agreement to fp32 rounding of the
        # optimizer, amplified by a few steps of training
        # (observed ~1e-3; the bound leaves room for the run-to-run spread that fp32 atomics order + Adam produce)
        assert abs(float(loss) - float(ref_loss)) < 1e-2, (step, float(loss), float(ref_loss))
    assert all(p.grad is not None for n, p in model.named_parameters() if n != "backbone.logit_scale")
    assert model.backbone.logit_scale.grad is None                  # unused in the reference too
    assert int(model.backbone.visual.bn1.num_batches_tracked) == 4


def _run_module(graph, steps=5, shapes=((4, 64),)):
    import os
    os.environ["CRIS_MODULE_GRAPH"] = "1" if graph else "0"
    try:
        dev = torch.device("cuda:0")
        model, groups = build_segmenter(NS(**TINY))
        clip, head = arch.specs_by_name("tiny")
        model.load_state_dict(arch.synthetic_state_dict(clip, head, 0))
        model = model.to(dev).train()
        opt = torch.optim.Adam(groups, lr=1e-4, weight_decay=0.0)
        scaler = torch.amp.GradScaler("cuda")
        losses = []
        for step in range(steps):
            b, s_ = shapes[step % len(shapes)]
            img, word, mask = (t.to(dev) for t in synth.make_batch(b, s_, 9, 0, step))
            with torch.autocast("cuda"):
                pred, target, loss = model(img, word, mask)
            opt.zero_grad()
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            losses.append((float(loss), float(pred.float().abs().sum())))
        torch.cuda.synchronize()
        return losses, {k: v.detach().clone() for k, v in model.state_dict().items()}, model
    finally:
        os.environ.pop("CRIS_MODULE_GRAPH", None)


@pytest.mark.parametrize("replay", ["graph", "cmdlist"])
def test_module_graph_replay_equals_the_eager_schedule(replay, monkeypatch):
    """the drop-in module replays two captured HIP graphs per step (forward + loss, backward + gradient export) from its
    autograd node; the dropout seed and GradScaler's factor live in device memory.  Same kernels in the same order: losses,
    logits and the parameters after five torch-Adam steps are bit-identical to the eager schedule - with one input shape
    (steps 0 eager, 1 capture, 2.. replay) and with two alternating shapes (two graph pairs over separate pools)."""
    monkeypatch.setenv("CRIS_MODULE_REPLAY", replay)          # cmdlist: host command lists instead of HIP graphs (same contract)
    for shapes in (((4, 64),), ((4, 64), (2, 96))):
        steps = 5 if len(shapes) == 1 else 8
        lg, sg, mg = _run_module(True, steps, shapes)
        le, se, _ = _run_module(False, steps, shapes)
        assert mg.graph_error is None and len(mg._graphs) == len(shapes), (mg.graph_error, len(mg._graphs))
        assert lg == le, (lg, le)
        assert all(torch.equal(sg[k], se[k]) for k in sg)


def _accumulate(direct, zero_style, steps=4):
    """two micro-batches per optimizer step (backward twice, then step), the gradient handed over directly or through autograd"""
    import os
    os.environ["CRIS_MODULE_DIRECT_GRAD"] = "1" if direct else "0"
    try:
        dev = torch.device("cuda:0")
        model, groups = build_segmenter(NS(**TINY))
        clip, head = arch.specs_by_name("tiny")
        model.load_state_dict(arch.synthetic_state_dict(clip, head, 0))
        model = model.to(dev).train()
        opt = torch.optim.Adam(groups, lr=1e-4, weight_decay=0.0)
        out = []
        for step in range(steps):
            if zero_style == "none":
                opt.zero_grad(set_to_none=True)
            elif zero_style == "zeros":
                opt.zero_grad(set_to_none=False)
            for micro in range(2):
                img, word, mask = _batch(2 * step + micro, dev)
                _, _, loss = model(img, word, mask)
                (0.5 * loss).backward()
            g = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
            opt.step()
            if zero_style == "never":                  # accumulate over everything: only the first step's gradients are compared
                out.append(g)
                break
            out.append(g)
        torch.cuda.synchronize()
        return out
    finally:
        os.environ.pop("CRIS_MODULE_DIRECT_GRAD", None)


@pytest.mark.parametrize("zero_style", ["none", "zeros", "never"])
def test_module_gradient_accumulation_direct_equals_autograd(zero_style):
    """with the gradients handed to `.grad` directly (single process default) a second backward without zero_grad
    must ADD to the first one, whatever zero_grad style the loop uses - the engine's buffers are re-used by every forward.
    Reference: autograd's own accumulation (CRIS_MODULE_DIRECT_GRAD=0)."""
    a, b = _accumulate(True, zero_style), _accumulate(False, zero_style)
    assert len(a) == len(b)
    for ga, gb in zip(a, b):
        assert ga.keys() == gb.keys()
        for k in ga:
            assert torch.equal(ga[k], gb[k]), k


def test_module_backward_of_a_superseded_forward_is_refused():
    """the engine keeps one step's activations: backward of a forward that a newer training forward has overwritten raises"""
    dev = torch.device("cuda:0")
    model, _ = build_segmenter(NS(**TINY))
    clip, head = arch.specs_by_name("tiny")
    model.load_state_dict(arch.synthetic_state_dict(clip, head, 0))
    model = model.to(dev).train()
    _, _, l0 = model(*_batch(0, dev))
    _, _, l1 = model(*_batch(1, dev))
    with pytest.raises(RuntimeError, match="NEWER training forward"):
        l0.backward()
    l1.backward()                                       # the latest one is fine
    kept = float(l1)
    _, _, l2 = model(*_batch(2, dev))
    assert float(l1) == kept                            # a loss value kept from a previous step does not change under the caller


def _loop(opt_cls, steps=5, scaler=True, seed_state=None):
    dev = torch.device("cuda:0")
    model, groups = build_segmenter(NS(**TINY))
    clip, head = arch.specs_by_name("tiny")
    model.load_state_dict(arch.synthetic_state_dict(clip, head, 0))
    model = model.to(dev).train()
    opt = opt_cls(groups, lr=1e-4, weight_decay=0.0)
    sc = torch.amp.GradScaler("cuda") if scaler else None
    losses = []
    for step in range(steps):
        img, word, mask = _batch(step, dev)
        with torch.autocast("cuda"):
            _, _, loss = model(img, word, mask)
        opt.zero_grad()
        if sc is not None:
            sc.scale(loss).backward()
            sc.step(opt)
            sc.update()
        else:
            loss.backward()
            opt.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    return losses, model, opt, sc


def test_optional_fused_adam_matches_torch_adam_under_the_reference_loop():
    """cris.pytorch_amd.optim.Adam in place of torch.optim.Adam (the one-line change at train.py:105), everything else of the
    loop unchanged (fp16 autocast, GradScaler): same losses as torch's Adam up to the optimizer's fp32 rounding; the fused path
    really ran (gradients are arena views, operand copies current after the step, no re-pack), the state_dict has torch's
    format and round-trips through torch.optim.Adam"""
    from cris.pytorch_amd import optim
    la, ma, oa, _ = _loop(optim.Adam)
    lb, mb, ob, _ = _loop(torch.optim.Adam)
    assert oa._usable() and ma._grad_views_active and ma._unpack is None
    assert ma._engine.packs_current                                   # the update rewrote the bf16 operand copies
    w = "neck.f2_v_proj.0.weight"                                      # a 3x3 convolution: `.grad` is a strided view of the arena
    g = dict(ma.named_parameters())[w].grad
    assert not g.is_contiguous() and g.shape == dict(ma.named_parameters())[w].shape
    for a, b in zip(la, lb):
        assert abs(a - b) < 1e-2, (la, lb)
    sd = oa.state_dict()
    assert sd["state"][0]["step"] == 5.0 and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    ob.load_state_dict(sd)                                            # torch's Adam takes it
    oa.load_state_dict(ob.state_dict())                               # and back
    assert int(oa._step_dev.item()) == 5
    pa, pb = dict(ma.named_parameters()), dict(mb.named_parameters())
    worst = max(float((pa[k] - pb[k]).abs().max()) for k in pa)
    assert worst < 2e-3, worst                                        # five steps of lr 1e-4: the two runs took the same steps


def _frozen_backbone_loop(opt_cls, steps=3):
    dev = torch.device("cuda:0")
    model, groups = build_segmenter(NS(**TINY))
    clip, head = arch.specs_by_name("tiny")
    model.load_state_dict(arch.synthetic_state_dict(clip, head, 0))
    model = model.to(dev).train()
    for p in model.backbone.parameters():                            # fine-tuning the head: the whole CLIP backbone frozen,
        p.requires_grad_(False)                                      # every parameter still handed to the optimizer (train.py:105)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    opt = opt_cls(groups, lr=1e-4, weight_decay=0.0)
    sc = torch.amp.GradScaler("cuda")
    for step in range(steps):
        img, word, mask = _batch(step, dev)
        with torch.autocast("cuda"):
            _, _, loss = model(img, word, mask)
        opt.zero_grad()
        sc.scale(loss).backward()
        sc.step(opt)
        sc.update()
    torch.cuda.synchronize()
    return model, before, float(sc.get_scale())


def test_fused_adam_with_frozen_parameters_falls_back_without_tripping_gradscaler():
    """round-5 advisor finding: GradScaler reads `_step_supports_amp_scaling` as True, skips its own unscale_ and leaves
    grad_scale / found_inf on the optimizer - and a step that THEN falls back to torch's Adam (a `.grad` is None: frozen
    parameters) hit an assertion inside torch (or would have stepped with scaled gradients).  The fallback now unscales itself:
    frozen parameters stay bit-identical, the others take the steps torch.optim.Adam takes."""
    from cris.pytorch_amd import optim
    ma, before, scale_a = _frozen_backbone_loop(optim.Adam)
    mb, _, scale_b = _frozen_backbone_loop(torch.optim.Adam)
    assert scale_a == scale_b
    pa, pb = dict(ma.named_parameters()), dict(mb.named_parameters())
    moved = 0
    for k, v in pa.items():
        if k.startswith("backbone."):
            assert torch.equal(v, before[k]), k                      # frozen: untouched
            assert v.grad is None
        else:
            assert float((v - pb[k]).abs().max()) < 1e-3, k          # three steps of lr 1e-4, the same steps as torch's Adam
            moved += int(not torch.equal(v, before[k]))
    assert moved > 50


def _accumulate_params(opt_cls, zero_style, steps=3):
    """two micro-batches per optimizer step under `opt_cls`; returns the parameters after `steps` optimizer steps"""
    dev = torch.device("cuda:0")
    model, groups = build_segmenter(NS(**TINY))
    clip, head = arch.specs_by_name("tiny")
    model.load_state_dict(arch.synthetic_state_dict(clip, head, 0))
    model = model.to(dev).train()
    opt = opt_cls(groups, lr=1e-4, weight_decay=0.0)
    for step in range(steps):
        if zero_style == "none":
            opt.zero_grad(set_to_none=True)
        elif zero_style == "zeros":
            opt.zero_grad(set_to_none=False)
        for micro in range(2):
            img, word, mask = _batch(2 * step + micro, dev)
            _, _, loss = model(img, word, mask)
            (0.5 * loss).backward()
        opt.step()
    torch.cuda.synchronize()
    return {n: p.detach().clone() for n, p in model.named_parameters()}, model, opt


@pytest.mark.parametrize("zero_style", ["none", "zeros", "never"])
def test_fused_adam_applies_the_accumulated_gradient(zero_style):
    """the fused update reads the gradient ARENA, which after a second backward without zero_grad (or with
    zero_grad(set_to_none=False)) holds the LAST micro-batch only while `.grad` holds the sum.  optim.Adam.step() copies such
    `.grad`s into their arena views first: the parameters after three optimizer steps of two micro-batches each equal
    torch.optim.Adam's, for every zero_grad style."""
    from cris.pytorch_amd import optim
    pa, ma, oa = _accumulate_params(optim.Adam, zero_style)
    pb, _, _ = _accumulate_params(torch.optim.Adam, zero_style)
    assert oa._usable() and ma._grad_views_active and oa._tab is not None        # the fused path ran
    # after the step the arena views hold exactly what `.grad` holds (the mechanism) ...
    for p_, v in zip(ma._step_params, ma._step_grads):
        assert p_.grad is v or torch.equal(p_.grad, v)
    # ... and the parameters took torch.optim.Adam's steps (the outcome).  Adam's first steps are sign-like (+-lr whatever the
    # gradient's size), so a step taken with only the LAST micro-batch's gradient moves a large share of the elements the other
    # way (by up to 2 lr per step); with equal gradients only elements whose gradient is rounding noise can differ (analytically
    # zero gradients: their sign is decided by the optimizers' own fp32 rounding) - the worst element may differ by 3 lr, but
    # the SHARE of elements off by more than lr / 2 stays tiny
    diff = torch.cat([(pa[k] - pb[k]).abs().flatten() for k in pa])
    share = float((diff > 5e-5).float().mean())
    print("accumulation %s: worst %.2e, share of elements off by > 5e-5: %.4f" % (zero_style, float(diff.max()), share))
    assert share < 0.02, share


def test_fused_adam_fallback_keeps_the_operand_copies_current():
    """(a) an optimizer whose hyperparameters rule the fused update out from the start (per-group weight
    decay) must not switch the module to gradient-view mode; (b) one that becomes ineligible later falls back to torch's step
    and marks the bf16 operand copies stale, so the next replayed forward re-packs them.  Checked on the copies themselves."""
    from cris.pytorch_amd import optim
    dev = torch.device("cuda:0")

    def fresh():
        model, groups = build_segmenter(NS(**TINY))
        clip, head = arch.specs_by_name("tiny")
        model.load_state_dict(arch.synthetic_state_dict(clip, head, 0))
        return model.to(dev).train(), groups

    def steps(model, opt, n, first=0):
        for step in range(first, first + n):
            img, word, mask = _batch(step, dev)
            _, _, loss = model(img, word, mask)
            opt.zero_grad()
            loss.backward()
            opt.step()

    def copies_current(model):
        e, w = model._engine, "neck.f2_cat.0.weight"                 # a 1x1 convolution: forward copy = bf16 of [N][Cin]
        p = dict(model.named_parameters())[w].detach()
        N, Cin = p.shape[0], p.shape[1]
        return torch.equal(e.WF[w].view(N, -1)[:, :Cin].float(), p.view(N, Cin).bfloat16().float())

    # (a) never eligible
    model, groups = fresh()
    groups[1]["weight_decay"] = 0.01
    opt = optim.Adam(groups, lr=1e-4, weight_decay=0.0)
    assert not model._grad_views
    steps(model, opt, 4)                                             # steps 2 .. 3 are replays
    assert not opt._usable() and not model._grad_views_active
    _ = model(*_batch(9, dev))                                       # a forward after the last update: re-packs in its replay
    assert copies_current(model)
    # (b) eligible, then not
    model, groups = fresh()
    opt = optim.Adam(groups, lr=1e-4, weight_decay=0.0)
    steps(model, opt, 3)
    assert opt._usable() and model._engine.packs_current and copies_current(model)
    opt.param_groups[1]["weight_decay"] = 0.01
    steps(model, opt, 2, first=3)
    assert not opt._usable() and not model._engine.packs_current      # torch's step ran: the copies are marked stale ...
    _ = model(*_batch(9, dev))
    assert copies_current(model)                                     # ... and the next forward repaired them
    opt.param_groups[1]["weight_decay"] = 0.0
    steps(model, opt, 2, first=5)                                    # eligible again: fused steps continue from torch's state
    assert opt._usable() and copies_current(model)
    assert int(opt._step_dev.item()) == 7


def test_fused_adam_under_distributed_data_parallel_like_train_py():
    """the reference's train.py ALWAYS initialises a process group and wraps the model in
    DistributedDataParallel (train.py:80-102), also on one GPU.  The fused update runs there too: `.grad` is then DDP's
    (averaged) gradient in a tensor of its own, which step() copies into the arena views.  One rank, gloo."""
    import os
    import tempfile
    import torch.distributed as dist
    from cris.pytorch_amd import optim
    assert not dist.is_initialized()
    dev = torch.device("cuda:0")
    res = {}
    with tempfile.TemporaryDirectory() as td:
        dist.init_process_group("gloo", init_method="file://" + os.path.join(td, "pg"), rank=0, world_size=1)
        try:
            for name, cls in (("cris", optim.Adam), ("torch", torch.optim.Adam)):
                model, groups = build_segmenter(NS(**TINY))
                clip, head = arch.specs_by_name("tiny")
                model.load_state_dict(arch.synthetic_state_dict(clip, head, 0))
                model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)                                   # train.py:97-98
                ddp = torch.nn.parallel.DistributedDataParallel(model.to(dev), device_ids=[0], find_unused_parameters=True)
                opt = cls(groups, lr=1e-4, weight_decay=0.0)
                sc = torch.amp.GradScaler("cuda")
                ddp.train()
                losses = []
                for step in range(5):
                    img, word, mask = _batch(step, dev)
                    with torch.autocast("cuda"):
                        _, _, loss = ddp(img, word, mask)
                    opt.zero_grad()
                    sc.scale(loss).backward()
                    sc.step(opt)
                    sc.update()
                    losses.append(float(loss))
                res[name] = (losses, {n: p.detach().clone() for n, p in model.named_parameters()}, opt, model)
        finally:
            dist.destroy_process_group()
    la, pa, oa, ma = res["cris"]
    lb, pb, _, _ = res["torch"]
    assert oa._usable() and oa._tab is not None and ma._grad_views_active and int(oa._step_dev.item()) == 5
    for a, b in zip(la, lb):
        assert abs(a - b) < 1e-2, (la, lb)
    assert max(float((pa[k] - pb[k]).abs().max()) for k in pa) < 2e-3


def test_module_notices_a_replaced_middle_parameter():
    """the per-step fast key looked at the first and the last parameter only - replacing a MIDDLE parameter after
    the first forward (weight surgery, a re-initialised head) kept the engine training the old tensor through its raw pointer.
    Now any parameter / module registration anywhere bumps an epoch the key contains, and the key sums every data_ptr: both a
    newly assigned nn.Parameter and a re-pointed `.data` rebuild the engine on the next forward."""
    dev = torch.device("cuda:0")
    model, _ = build_segmenter(NS(**TINY))
    clip, head = arch.specs_by_name("tiny")
    model.load_state_dict(arch.synthetic_state_dict(clip, head, 0))
    model = model.to(dev).train()
    def fwd():
        model._steps = 0                       # (the dropout seed follows the step count: the same masks for every comparison)
        return model(*_batch(0, dev))[2]
    l0 = fwd()
    name = "neck.f2_cat.0.weight"
    holder = model.neck.f2_cat[0]
    old = holder.weight
    assert model._engine.P[name].data_ptr() == old.data_ptr()
    holder.weight = torch.nn.Parameter(torch.zeros_like(old))          # (1) a new Parameter object in the middle of the tree
    l1 = fwd()
    assert model._engine.P[name].data_ptr() == holder.weight.data_ptr() != old.data_ptr()
    assert float(l1) != float(l0)                                        # the zeroed layer is what ran
    holder.weight.data = old.data.clone()                                # (2) `.data` re-pointed: no registration, new storage
    l2 = fwd()
    assert model._engine.P[name].data_ptr() == holder.weight.data_ptr()
    assert float(l2) == float(l0)                                        # the original values again: the original loss, bit for bit


def test_optional_fused_adam_skips_the_step_on_found_inf():
    """GradScaler's contract for `_step_supports_amp_scaling` optimizers: found_inf != 0 -> nothing changes, the step count
    does not advance; the gradients are divided by grad_scale inside the update"""
    from cris.pytorch_amd import optim
    _, model, opt, _ = _loop(optim.Adam, steps=2, scaler=False)
    dev = torch.device("cuda:0")
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    step0 = int(opt._step_dev.item())
    img, word, mask = _batch(7, dev)
    _, _, loss = model(img, word, mask)
    opt.zero_grad()
    (loss * 1024.0).backward()
    opt.grad_scale, opt.found_inf = torch.full((), 1024.0, device=dev), torch.ones((), device=dev)
    opt.step()
    assert int(opt._step_dev.item()) == step0
    after = model.state_dict()
    assert all(torch.equal(before[k], after[k]) for k in before if "num_batches_tracked" not in k and "running" not in k)
    opt.found_inf = torch.zeros((), device=dev)
    opt.step()                                                        # now it is taken, with the gradients divided by 1024
    assert int(opt._step_dev.item()) == step0 + 1
    moved = max(float((before[k] - after[k]).abs().max()) for k in before if k.endswith("weight") and "bn" not in k)
    assert 0 < moved < 5e-4                                           # |Adam step| <= lr = 1e-4 (not 1024x that)


def test_module_eval_and_checkpoint_reload():
    dev = torch.device("cuda:0")
    model, _ = build_segmenter(NS(**TINY))
    model = model.to(dev)
    img, word, mask = _batch(0, dev)
    model.train()
    model(img, word, mask)[2].backward()
    model.eval()
    p1 = model(img, word)
    assert p1.shape == (4, 1, 16, 16) and not p1.requires_grad
    # checkpoint interchange: a fresh module loaded from the state_dict gives the same eval output
    other, _ = build_segmenter(NS(**TINY))
    other.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    other = other.to(dev).eval()
    p2 = other(img, word)
    assert torch.equal(p1, p2)


def test_launch_modes_agree():
    """eager Python schedule, HIP-graph replay and host command-list replay run the same kernels with the same device-side
    step counter / dropout seed: their loss curves agree (fp32 atomics order is the only difference)."""
    dev = torch.device("cuda:0")
    clip, head = arch.specs_by_name("tiny")
    sd = arch.synthetic_state_dict(clip, head, 0)
    curves = {}
    for mode in ("eager", "graph", "cmdlist"):
        tr = NativeTrainer(clip, head, sd, dev, base_lr=1e-4, launch=mode)
        out = []
        for step in range(6):
            img, word, mask = _batch(step, dev)
            loss, metric = tr.train_step(img, word, mask)
            out.append(float(loss))
        assert tr.launch == mode and tr.graph_error is None
        assert (tr._graph is not None) == (mode == "graph") and (tr._cmds is not None) == (mode == "cmdlist")
        curves[mode] = out
    print(curves)
    for mode in ("graph", "cmdlist"):
        d = max(abs(a - b) for a, b in zip(curves[mode], curves["eager"]))
        assert d < 1e-2, (mode, curves)             # observed <= 1.5e-3


def test_native_trainer_checkpoint_resume_is_exact_and_torch_compatible():
    """train.py:159-174,192-207 (--resume): model + optimizer state out of the native trainer, into a fresh one - the
    continued run is bit-identical to the uninterrupted one (the path is deterministic) - and the optimizer part loads into
    the torch.optim.Adam the reference builds from build_segmenter's param_list."""
    dev = torch.device("cuda:0")
    clip, head = arch.specs_by_name("tiny")
    sd = arch.synthetic_state_dict(clip, head, 0)
    a = NativeTrainer(clip, head, sd, dev, base_lr=1e-4, launch="eager")
    for step in range(3):
        a.train_step(*_batch(step, dev))
    msd, osd = a.model_state_dict(), a.optimizer_state_dict()
    cont = [float(a.train_step(*_batch(step, dev))[0]) for step in (3, 4)]
    b = NativeTrainer(clip, head, msd, dev, base_lr=1e-4, launch="eager")
    b.load_optimizer_state_dict(osd)
    resumed = [float(b.train_step(*_batch(step, dev))[0]) for step in (3, 4)]
    assert resumed == cont, (resumed, cont)
    assert all(torch.equal(a.engine.P[k], b.engine.P[k]) for k in a.engine.P)
    # torch-format compatibility with the reference's optimizer construction (train.py:105-107)
    model, groups = build_segmenter(NS(**TINY))
    opt = torch.optim.Adam(groups, lr=1e-4, weight_decay=0.0)
    opt.load_state_dict(osd)
    n_params = sum(len(g["params"]) for g in opt.state_dict()["param_groups"])
    assert n_params == len(list(model.parameters()))
    names = [n for n, _ in model.named_parameters()]
    i = osd["param_groups"][1]["params"][0]                   # first parameter of the head group
    first_head = [n for n in names if not (n.startswith("backbone") and "positional_embedding" not in n)][0]
    assert tuple(osd["state"][i]["exp_avg"].shape) == tuple(dict(model.named_parameters())[first_head].shape)
    assert int(msd["backbone.visual.bn1.num_batches_tracked"]) == 3


def test_module_under_dataparallel_like_test_py():
    """test.py:70-72: `model = torch.nn.DataParallel(model).cuda()` then eval forward.  On one visible GPU DataParallel calls
    the wrapped module directly (no replicate): outputs equal the bare module's; the `module.` key prefix test.py:74-78 strips
    / expects round-trips."""
    dev = torch.device("cuda:0")
    model, _ = build_segmenter(NS(**TINY))
    wrapped = torch.nn.DataParallel(model, device_ids=[0]).cuda()
    img, word, _ = _batch(0, dev)
    wrapped.eval()
    p1 = wrapped(img, word)
    p2 = model(img, word)
    assert p1.shape == (4, 1, 16, 16) and torch.equal(p1, p2)
    sd = wrapped.state_dict()
    assert all(k.startswith("module.") for k in sd)
    other, _ = build_segmenter(NS(**TINY))
    other = torch.nn.DataParallel(other, device_ids=[0]).cuda()
    other.load_state_dict(sd, strict=True)                      # test.py:78
    other.eval()
    assert torch.equal(other(img, word), p1)


def test_module_eval_runs_folded_and_tracks_parameter_changes(monkeypatch):
    """`model.eval()` forwards run on the inference engine (BatchNorms folded, HIP graph per shape; infer.py).  The folded
    weights are a cache: an optimizer step (+ the running statistics the training forward updates) and a load_state_dict must
    each refresh it - checked against freshly built modules."""
    dev = torch.device("cuda:0")
    clip, head = arch.specs_by_name("tiny")
    sd0 = arch.synthetic_state_dict(clip, head, 0)
    model, groups = build_segmenter(NS(**TINY))
    model.load_state_dict(sd0)
    model = model.to(dev).eval()
    img, word, mask = _batch(0, dev)
    a = [model(img, word) for _ in range(3)]                      # eager, captured, replayed
    assert torch.equal(a[0], a[1]) and torch.equal(a[0], a[2]) and a[0].data_ptr() != a[1].data_ptr()
    run = model._infer
    assert len(run.engine._fold) == len(run.engine.bn_prefixes) - (3 if run.engine.state_f32 else 2) and next(iter(run._shapes.values()))["graph"] is not None
    monkeypatch.setenv("CRIS_EVAL_FOLD", "0")
    b = model(img, word)                                          # the training engine's eval forward (apply kernels)
    monkeypatch.delenv("CRIS_EVAL_FOLD")
    err = float((a[0] - b).norm() / b.norm())
    assert 0 < err < 3e-2, err
    # one optimizer step: parameters AND running statistics move
    model.train()
    opt = torch.optim.Adam(groups, lr=1e-3)
    model(img, word, mask)[2].backward()
    opt.step()
    model.eval()
    c = model(img, word)
    fresh, _ = build_segmenter(NS(**TINY))
    fresh.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    d = fresh.to(dev).eval()(img, word)
    assert torch.equal(c, d) and not torch.equal(c, a[0])
    assert model._infer is run                                    # same runner, refreshed
    # back to the first state through load_state_dict
    model.load_state_dict(sd0)
    assert torch.equal(model(img, word), a[0])
