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
