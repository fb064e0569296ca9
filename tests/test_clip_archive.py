"""Checkpoint interchange at the CLIP-archive boundary (reference model/segmenter.py:14-16,
model/clip.py:503-554): `CRIS(cfg)` must accept a REAL TorchScript archive through `torch.jit.load(path).state_dict()`,
infer the architecture from its keys, drop the three scalar entries, leave `attnpool.connect.*` at its fresh initialisation
(strict=False) and round conv / linear / attention-projection tensors through fp16.  RN50.pt is not available offline, so the
archive is a scripted module tree carrying the tiny synthetic CLIP tensors under the same key names."""
import os
from types import SimpleNamespace as NS

import torch
from torch import nn

from cris.pytorch_amd import arch
from cris.pytorch_amd.model import CRIS

TINY = dict(word_len=9, fpn_in=[128, 256, 128], fpn_out=[64, 128, 256], num_layers=2, vis_dim=128, num_head=2,
            dim_ffn=256, dropout=0.1, intermediate=False, word_dim=128, base_lr=1e-4, lr_multi=0.1)


class _Node(nn.Module):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x


def _scripted_archive(sd):
    root = _Node()
    for key, value in sd.items():
        *path, leaf = key.split(".")
        m = root
        for p in path:
            if p not in m._modules:
                m.add_module(p, _Node())
            m = m._modules[p]
        if leaf in ("running_mean", "running_var", "num_batches_tracked") or not value.is_floating_point():
            m.register_buffer(leaf, value.clone())
        else:
            m.register_parameter(leaf, nn.Parameter(value.clone()))
    return torch.jit.script(root)


def test_cris_loads_a_torchscript_clip_archive(tmp_path):
    clip, head = arch.specs_by_name("tiny")
    sd = dict(arch.clip_state_dict_view(arch.synthetic_state_dict(clip, head, 3)))
    sd = {k: v for k, v in sd.items() if "attnpool.connect" not in k}          # not part of an OpenAI archive (clip.py:76-78)
    with torch.no_grad():                                                       # make fp16 rounding observable
        sd["visual.conv1.weight"] = sd["visual.conv1.weight"] + 1e-4
    sd["input_resolution"] = torch.tensor(64)
    sd["context_length"] = torch.tensor(clip.context_length)
    sd["vocab_size"] = torch.tensor(clip.vocab_size)
    path = os.path.join(tmp_path, "tiny_clip.pt")
    torch.jit.save(_scripted_archive(sd), path)

    model = CRIS(NS(clip_pretrain=path, **TINY))
    assert model.clip_spec == clip                                              # architecture inferred from the keys
    own = model.backbone.state_dict()
    assert not any(k in own for k in ("input_resolution", "context_length", "vocab_size"))
    w = sd["visual.conv1.weight"]
    assert torch.equal(own["visual.conv1.weight"], w.half().float()) and not torch.equal(w, w.half().float())
    for k in ("visual.bn1.weight", "visual.bn1.running_var", "positional_embedding", "ln_final.weight",
              "visual.attnpool.positional_embedding", "token_embedding.weight"):
        assert torch.equal(own[k], sd[k]), k                                    # norms / embeddings load unrounded
    for k in ("transformer.resblocks.0.attn.in_proj_weight", "transformer.resblocks.1.mlp.c_fc.bias", "text_projection",
              "visual.attnpool.q_proj.weight"):
        assert torch.equal(own[k], sd[k].half().float()), k
    assert any("attnpool.connect" in k for k in own)                            # new, freshly initialised block stays
    # and the module's own checkpoint round-trips through a plain file like train.py:159-166 / test.py:74-78 do
    ck = os.path.join(tmp_path, "last_model.pth")
    torch.save({"state_dict": model.state_dict()}, ck)
    other = CRIS(NS(clip_pretrain="synthetic:tiny", **TINY))
    res = other.load_state_dict(torch.load(ck, map_location="cpu")["state_dict"], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert all(torch.equal(a, b) for a, b in zip(other.state_dict().values(), model.state_dict().values()))
