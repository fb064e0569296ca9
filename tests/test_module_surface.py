"""CPU checks of the drop-in surface: `from model import build_segmenter` / `CRIS(cfg)` keep the reference's
parameter names, shapes and optimizer grouping; the module survives convert_sync_batchnorm and a state_dict round trip;
and it FAILS LOUDLY off the GPU (no CPU fallback in the product path)."""
import os
from types import SimpleNamespace as NS

import pytest
import torch
from torch import nn

from conftest import GOLDEN
from cris.pytorch_amd.model import CRIS, build_segmenter

R50 = dict(clip_pretrain="synthetic", word_len=17, fpn_in=[512, 1024, 1024], fpn_out=[256, 512, 1024], num_layers=3,
           vis_dim=512, num_head=8, dim_ffn=2048, dropout=0.1, intermediate=False, word_dim=1024, base_lr=1e-4, lr_multi=0.1)
TINY = dict(clip_pretrain="synthetic:tiny", word_len=9, fpn_in=[128, 256, 128], fpn_out=[64, 128, 256], num_layers=2,
            vis_dim=128, num_head=2, dim_ffn=256, dropout=0.1, intermediate=False, word_dim=128, base_lr=1e-4, lr_multi=0.1)


def test_r50_keys_and_param_groups_match_reference():
    model, groups = build_segmenter(NS(**R50))
    ref = [l.split(" ") for l in open(os.path.join(GOLDEN, "state_dict_keys_r50.txt")).read().split("\n") if l]
    sd = model.state_dict()
    assert list(sd.keys()) == [r[0] for r in ref]                 # names AND order of the reference module
    assert sum(p.numel() for p in model.parameters()) == 146849122
    assert len(groups[0]["params"]) == 325 and len(groups[1]["params"]) == 124     # SURVEY.md a12 [probe]
    assert groups[0]["initial_lr"] == pytest.approx(1e-5) and groups[1]["initial_lr"] == pytest.approx(1e-4)
    # (round 6: the groups also ask torch.optim.Adam for its fused implementation - same class, same constructor call as train.py:105)
    assert all(set(g) == {"params", "initial_lr", "fused"} and g["fused"] is True for g in groups)
    names0 = {id(p) for p in groups[0]["params"]}
    for k, p in model.named_parameters():
        expect0 = k.startswith("backbone") and "positional_embedding" not in k
        assert (id(p) in names0) == expect0, k


def test_clip_weights_are_fp16_rounded_like_the_reference_loader():
    model = CRIS(NS(**TINY))
    w = model.backbone.visual.layer1[0].conv1.weight.data
    assert torch.equal(w, w.half().float())                      # conv weights pass through fp16 (clip.py:552)
    g = model.backbone.visual.bn1.weight.data
    assert not torch.equal(g, g.half().float())                  # BatchNorm parameters do not


def test_sync_batchnorm_conversion_and_state_dict_roundtrip():
    model = CRIS(NS(**TINY))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    conv = nn.SyncBatchNorm.convert_sync_batchnorm(model)
    assert any(isinstance(m, nn.SyncBatchNorm) for m in conv.modules())
    assert list(conv.state_dict().keys()) == list(sd.keys())
    other = CRIS(NS(**TINY))
    with torch.no_grad():
        for p in other.parameters():
            p.add_(1.0)
    missing = other.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    for k, v in other.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_no_cpu_fallback():
    model = CRIS(NS(**TINY))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.zeros(1, 3, 64, 64), torch.zeros(1, 9, dtype=torch.long))


def test_optional_optimizer_binds_only_when_the_fused_update_can_run():
    """cris.pytorch_amd.optim.Adam (round-4 advisor findings): the module is switched to gradient-view mode - where its replayed
    forward carries no re-pack of the bf16 operand copies - ONLY by an optimizer that holds every gradient parameter of the module with
    hyperparameters the fused update supports; anything else leaves the module alone and is plain torch.optim.Adam."""
    from cris.pytorch_amd import optim
    model, groups = build_segmenter(NS(**TINY))
    opt = optim.Adam(groups, lr=1e-4, weight_decay=0.0)               # the reference's two groups, one weight decay: eligible
    assert opt._cris is model and model._grad_views and opt._hyper_ok()
    assert not opt._usable()                                          # (no engine yet: nothing ran on a GPU)

    model2, groups2 = build_segmenter(NS(**TINY))
    groups2[1]["weight_decay"] = 0.01                                 # per-group weight decay: torch's Adam does that, the fused one not
    opt2 = optim.Adam(groups2, lr=1e-4, weight_decay=0.0)
    assert opt2._cris is model2 and not model2._grad_views and not opt2._hyper_ok()

    model3, groups3 = build_segmenter(NS(**TINY))
    opt3 = optim.Adam([groups3[1]], lr=1e-4)                          # a SUBSET of the parameters (fine-tuning the head only)
    assert opt3._cris is None and not model3._grad_views

    other = nn.Linear(4, 4)
    opt4 = optim.Adam(other.parameters(), lr=1e-3)                    # parameters of something else: plain torch.optim.Adam
    other(torch.ones(2, 4)).sum().backward()
    before = other.weight.detach().clone()
    opt4.step()
    assert opt4._cris is None and not torch.equal(before, other.weight)

    model5, groups5 = build_segmenter(NS(**TINY))
    opt5 = optim.Adam(groups5, lr=1e-4, amsgrad=True)
    assert not model5._grad_views                                     # amsgrad: not the fused update's arithmetic


def test_fast_key_sees_a_replaced_middle_parameter():
    """the per-step key of the module -> engine binding (round-4 advisor finding: it looked at the first and last parameter only)"""
    model, _ = build_segmenter(NS(**TINY))
    k0 = model._fast_key("cuda:0")
    assert model._fast_key("cuda:0") == k0                            # stable while nothing changes
    holder = model.neck.f2_cat[0]
    old = holder.weight
    holder.weight = nn.Parameter(old.detach().clone())                # a new Parameter object in the middle of the tree
    k1 = model._fast_key("cuda:0")
    assert k1 != k0
    holder.weight.data = old.detach().clone()                         # `.data` re-pointed: no registration, new storage
    assert model._fast_key("cuda:0") != k1


def test_registrations_outside_the_tree_do_not_force_a_parameter_walk():
    """round-5 advisor finding: the registration hooks are process-global; a module built per step anywhere in the user's
    process (a loss or a metric with parameters / children) must not make every forward re-walk the 600-module tree"""
    from cris.pytorch_amd.model import segmenter as S
    model, _ = build_segmenter(NS(**TINY))
    k0 = model._fast_key("cuda:0")
    e0 = S._REGISTRATION_EPOCH[0]
    for _ in range(3):
        loss_mod = nn.Sequential(nn.Linear(3, 3), nn.BatchNorm1d(3))          # parameters, buffers and sub-modules: all foreign
        loss_mod.extra = nn.Parameter(torch.zeros(1))
    assert S._REGISTRATION_EPOCH[0] == e0 and model._fast_key("cuda:0") == k0
    plist = model._plist
    assert model._fast_key("cuda:0") == k0 and model._plist is plist        # no walk happened
    # a graft INTO the tree is seen, and so is a later registration on the grafted module
    model.neck.f2_cat[0] = nn.Conv2d(model.neck.f2_cat[0].in_channels, model.neck.f2_cat[0].out_channels, 1, bias=False)
    assert S._REGISTRATION_EPOCH[0] > e0
    k1 = model._fast_key("cuda:0")
    assert k1 != k0 and model._plist is not plist
    e1 = S._REGISTRATION_EPOCH[0]
    model.neck.f2_cat[0].weight = nn.Parameter(model.neck.f2_cat[0].weight.detach().clone())
    assert S._REGISTRATION_EPOCH[0] > e1 and model._fast_key("cuda:0") != k1


def test_ddp_ignore_list_names_every_parameter_but_one():
    """SURVEY.md 8e option B: what a DistributedDataParallel constructor reads from the module (it must keep one parameter
    that requires a gradient; buffers are left to DDP unless the BatchNorms are SyncBatchNorm)"""
    import os
    model, _ = build_segmenter(NS(**TINY))
    names = set(model._ddp_params_and_buffers_to_ignore)
    assert not model._ddp_asked                                           # read by a test, not by a DDP constructor
    pn = {n for n, _ in model.named_parameters()}
    assert pn - names == {"backbone.logit_scale"} and not (names - pn)      # plain BatchNorm: the buffers stay DDP's
    sync = nn.SyncBatchNorm.convert_sync_batchnorm(model)
    assert sync is model
    names = set(model._ddp_params_and_buffers_to_ignore)
    assert names == (pn - {"backbone.logit_scale"}) | {n for n, _ in model.named_buffers()}
    nn.parallel.DistributedDataParallel._set_params_and_buffers_to_ignore_for_model(model, ["backbone.logit_scale"])
    assert "backbone.logit_scale" in model._ddp_params_and_buffers_to_ignore
    os.environ["CRIS_DDP_SELF_EXCHANGE"] = "0"
    try:
        assert not hasattr(model, "_ddp_params_and_buffers_to_ignore")
    finally:
        del os.environ["CRIS_DDP_SELF_EXCHANGE"]


def test_ddp_constructor_leaves_the_parameters_to_the_module(tmp_path):
    """the wrap of train.py:100-102 on the CPU with a one-rank gloo group: DistributedDataParallel accepts the module with ONE
    managed parameter, the module knows it was asked by a DDP constructor and holds the wrapper (for its no_sync() state)"""
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="file://" + str(tmp_path / "pg"), rank=0, world_size=1)
    try:
        model, _ = build_segmenter(NS(**TINY))          # (plain BatchNorm: DDP refuses SyncBatchNorm modules on the CPU)
        ddp = nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)
        assert model._ddp_asked and model._ddp_ref() is ddp
        assert len(ddp._module_parameters) == 1 and ddp._module_parameters[0] is model.backbone.logit_scale
        assert len(list(ddp.parameters())) == len(list(model.parameters()))       # the optimizer still sees every parameter
        with ddp.no_sync():
            assert not model._ddp_ref().require_backward_grad_sync
        assert model._ddp_ref().require_backward_grad_sync
    finally:
        dist.destroy_process_group()
