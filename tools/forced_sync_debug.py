"""debug: first training-forward loss of the drop-in module (default-initialised head) with the SyncBN code paths forced on a
world of one, per exchange form, against the plain BatchNorm path - from identical parameters and running statistics"""
import os
import sys
import tempfile

import torch
import torch.distributed as dist
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from types import SimpleNamespace as NS          # noqa: E402
from cris.pytorch_amd import debug, synth        # noqa: E402
from cris.pytorch_amd.model import build_segmenter   # noqa: E402

spec = sys.argv[1] if len(sys.argv) > 1 else "r50"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
d = tempfile.mkdtemp()
dist.init_process_group("nccl", init_method="file://" + os.path.join(d, "pg"), rank=0, world_size=1, device_id=dev)
if spec == "tiny":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_module_surface import TINY
    cfg, B, S, L = NS(**TINY), 4, 64, 9
else:
    cfg = NS(clip_pretrain="synthetic", word_len=17, fpn_in=[512, 1024, 1024], fpn_out=[256, 512, 1024], num_layers=3, vis_dim=512,
             num_head=8, dim_ffn=2048, dropout=0.1, intermediate=False, word_dim=1024, base_lr=1e-4, lr_multi=0.1, sync_bn=True)
    B, S, L = 8, 416, 17
model, _ = build_segmenter(cfg)
model = nn.SyncBatchNorm.convert_sync_batchnorm(model).cuda().train()
sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
img, word, mask = (t.to(dev) for t in synth.make_batch(B, S, L, 0, 0))
os.environ["CRIS_MODULE_GRAPH"] = "0"


def run(force, p2p, fused):
    model.load_state_dict(sd0)
    debug.HOOKS.force_dist = force
    os.environ["CRIS_SYNCBN_P2P"] = "1" if p2p else "0"
    os.environ["CRIS_SYNCBN_FUSED"] = "1" if fused else "0"
    model._engine = None
    model._engine_fast_key = None
    with torch.autocast("cuda"):
        pred, _, loss = model(img, word, mask)
    torch.cuda.synchronize()
    rm = model.state_dict()["neck.f1_v_proj.1.running_mean"].double().abs().sum().item()
    rv = model.state_dict()["backbone.visual.layer4.0.bn3.running_var"].double().sum().item()
    return float(loss), float(pred.float().abs().sum()), rm, rv, getattr(model, "syncbn_exchange", None)


for name, args in (("plain", (False, False, False)), ("forced collective", (True, False, False)), ("forced p2p kernel", (True, True, False)),
                   ("forced p2p fused", (True, True, True)), ("plain again", (False, False, False))):
    print("%-20s loss %.6f |pred| %.3f rm %.6f rv %.6f  %s" % ((name,) + run(*args)), flush=True)
dist.destroy_process_group()
