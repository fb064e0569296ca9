"""The loop restatement of tests/ref_loop.py is pinned to the reference: the reference's OWN `engine.engine.train` (imported
from /root/reference with its logging dependencies stubbed) and `ref_loop.train_steps` drive two copies of one small CPU
model over the same batches with the same Adam / MultiStepLR / GradScaler objects; parameters and BatchNorm buffers must be
bit-identical afterwards, and trainMetricGPU must agree on random inputs.  Skipped where the reference is absent (GPU box)."""
import copy
import os
import sys
import time
import types
from types import SimpleNamespace as NS

import pytest
import torch
import torch.distributed as dist
from torch import nn

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
import ref_harness  # noqa: E402
import ref_loop  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_harness.reference_available(), reason="reference not present")


class _Toy(nn.Module):
    """the model contract of model/segmenter.py:29-62 in miniature: (img, word, mask) -> (pred.detach(), mask/4, loss)"""

    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 4, 3, stride=4, padding=1)
        self.bn = nn.BatchNorm2d(4)
        self.emb = nn.Embedding(50, 4)
        self.head = nn.Conv2d(4, 1, 1)

    def forward(self, img, word, mask=None):
        x = torch.relu(self.bn(self.conv(img))) * self.emb(word).mean(1)[:, :, None, None]
        pred = self.head(x)
        if self.training:
            mask = torch.nn.functional.interpolate(mask, pred.shape[-2:], mode="nearest").detach()
            loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, mask)
            return pred.detach(), mask, loss
        return pred.detach()


def _reference_train():
    ref_harness.import_reference()
    for name in ("tqdm",):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.tqdm = lambda x, *a, **k: x
            sys.modules[name] = m
    wb = sys.modules["wandb"]
    wb.log = lambda *a, **k: None
    import engine.engine as ref_engine           # the reference's engine/engine.py
    return ref_engine


@pytest.fixture(scope="module")
def pg():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
    yield
    dist.destroy_process_group()


def test_train_metric_matches_reference():
    ref_harness.import_reference()
    from utils.misc import trainMetricGPU
    g = torch.Generator().manual_seed(0)
    for _ in range(5):
        x = torch.randn(4, 1, 16, 16, generator=g) * 2
        t = (torch.rand(4, 1, 16, 16, generator=g) > 0.6).float()
        a = trainMetricGPU(x.clone(), t)
        b = ref_loop.train_metric_gpu(x.clone(), t)
        assert float(a[0]) == float(b[0]) and float(a[1]) == float(b[1])


@pytest.mark.parametrize("max_norm", [0.0, 0.5])
def test_restated_loop_equals_reference_engine_train(pg, monkeypatch, max_norm):
    ref_engine = _reference_train()
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)        # CPU run of a loop written for cuda
    monkeypatch.setattr(time, "sleep", lambda s: None)                             # engine/engine.py:30
    g = torch.Generator().manual_seed(1)
    batches = [(torch.randn(4, 3, 32, 32, generator=g), torch.randint(0, 50, (4, 5), generator=g),
                (torch.rand(4, 32, 32, generator=g) > 0.5).float()) for _ in range(5)]
    torch.manual_seed(0)
    m_ref = _Toy()
    m_own = copy.deepcopy(m_ref)

    def setup(m):
        groups = [{"params": [p for n, p in m.named_parameters() if n.startswith("conv")], "initial_lr": 1e-3},
                  {"params": [p for n, p in m.named_parameters() if not n.startswith("conv")], "initial_lr": 1e-2}]
        opt = torch.optim.Adam(groups, lr=1e-2, weight_decay=0.0)
        sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[35], gamma=0.1)
        return opt, sched, torch.amp.GradScaler("cuda", enabled=False)

    opt, sched, scaler = setup(m_ref)
    args = NS(epochs=1, max_norm=max_norm, print_freq=10 ** 9)
    ref_engine.train(batches, m_ref, opt, sched, scaler, 0, args)
    opt2, _, scaler2 = setup(m_own)
    res = ref_loop.train_steps(batches, m_own, opt2, scaler2, max_norm=max_norm, device_type="cpu")
    assert len(res) == len(batches)
    for (k, a), (_, b) in zip(m_ref.state_dict().items(), m_own.state_dict().items()):
        assert torch.equal(a, b), k
