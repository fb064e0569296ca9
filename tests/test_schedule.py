"""Learning-rate schedule of the native trainer == what the reference's optimizer/scheduler construction does
(train.py:105-110,210), pinned against torch's own Adam + MultiStepLR driven the same way."""
import warnings

import torch
from torch.optim.lr_scheduler import MultiStepLR

from cris.pytorch_amd.trainer import epoch_group_lrs


def _reference_lrs(base_lr, lr_multi, milestones, gamma, epochs):
    a, b = torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))
    groups = [{"params": [a], "initial_lr": lr_multi * base_lr}, {"params": [b], "initial_lr": base_lr}]   # build_segmenter
    opt = torch.optim.Adam(groups, lr=base_lr, weight_decay=0.0)
    sched = MultiStepLR(opt, milestones=milestones, gamma=gamma)
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for epoch in range(epochs):
            out.append((opt.param_groups[0]["lr"], opt.param_groups[1]["lr"]))
            opt.step()
            sched.step(epoch + 1)                       # train.py:143,210: scheduler.step(epoch_log), epoch_log = epoch + 1
    return out


def test_epoch_lrs_match_torch_multisteplr_driven_like_the_reference():
    for base_lr, lr_multi, ms, gamma, n in [(1e-4, 0.1, [35], 0.1, 50), (5e-5, 0.5, [3, 7], 0.2, 12)]:
        ref = _reference_lrs(base_lr, lr_multi, ms, gamma, n)
        for e, (rb, rh) in enumerate(ref):
            lb, lh = epoch_group_lrs(e, base_lr, lr_multi, ms, gamma)
            assert abs(lb - rb) <= 1e-12 * max(rb, 1e-30) + 1e-20, (e, lb, rb)
            assert abs(lh - rh) <= 1e-12 * max(rh, 1e-30) + 1e-20, (e, lh, rh)
    # the survey's probe: epoch 0 trains everything at base_lr, from epoch 1 the backbone runs at lr_multi*base_lr
    assert epoch_group_lrs(0, 1e-4, 0.1, [35], 0.1) == (1e-4, 1e-4)
    assert abs(epoch_group_lrs(1, 1e-4, 0.1, [35], 0.1)[0] - 1e-5) < 1e-18
