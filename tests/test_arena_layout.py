"""Gradient-arena invariants the staged data-parallel exchange relies on (cris/pytorch_amd/engine.py _build_grad_arena,
trainer.py on_stage -> comm.allreduce_async(grad_arena[lo:hi]); reference semantics: DDP all-reduces EVERY parameter
gradient once per step, train.py:100-102).  Pure host logic: the Engine is constructed on the CPU, no kernel runs."""
import pytest
import torch

from cris.pytorch_amd import arch
from cris.pytorch_amd.engine import Engine
from cris.pytorch_amd.trainer import split_state_dict


def _engine(spec):
    clip, head = arch.specs_by_name(spec)
    params, buffers = split_state_dict(arch.synthetic_state_dict(clip, head, 0), "cpu")
    return Engine(clip, head, params, buffers, torch.device("cpu")), params


@pytest.mark.parametrize("spec", ["tiny", "r50"])
def test_stage_ranges_tile_the_arena_and_hold_every_gradient_once(spec):
    e, params = _engine(spec)
    total = e.grad_arena.numel()
    ranges = [e.stage_ranges[s] for s in sorted(e.stage_ranges)]
    assert sorted(e.stage_ranges) == list(range(8))
    assert ranges[0][0] == 0 and ranges[-1][1] == total
    for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
        assert a1 == b0 and a0 < a1                              # contiguous, non-empty, in stage order
    base = e.grad_arena.data_ptr()
    covered = torch.zeros(total, dtype=torch.int32)
    for name, p in params.items():
        g = e.G[name]
        off = (g.data_ptr() - base) // 4
        assert (g.data_ptr() - base) % 16 == 0, name                # 16-byte aligned views (vector loads in Adam / unpack)
        assert g.numel() >= p.numel(), name                         # GEMM layouts may pad channels, never shrink
        lo, hi = e.stage_ranges[e.stage_of(name)]
        assert lo <= off and off + g.numel() <= hi, name            # inside the range exchanged for ITS stage
        covered[off:off + g.numel()] += 1
    assert int(covered.max()) == 1                                  # no two gradients share an element
    assert total - int(covered.sum()) < 4 * len(params)             # only alignment padding is uncovered
    if spec == "r50":
        assert sum(p.numel() for p in params.values()) == 146849122     # SURVEY.md 8c [probe]
        sizes_mb = [4 * (hi - lo) / 1e6 for lo, hi in ranges]
        assert 580 < sum(sizes_mb) < 600                               # 587 MB fp32 payload


def test_backward_finishes_stages_back_to_front():
    """Stage numbering is forward order (visual 0-3, text 4, neck 5, decoder 6, projector 7): backward completes 7, 6, 5
    first, so their (contiguous, arena-tail) ranges are on the wire while the encoders are still in backward."""
    e, params = _engine("tiny")
    of = e.stage_of
    assert of("proj.vis.0.0.weight") == 7 and of("decoder.layers.0.norm1.weight") == 6 and of("neck.f1_v_proj.0.weight") == 5
    assert of("backbone.token_embedding.weight") == 4 and of("backbone.positional_embedding") == 4
    assert of("backbone.visual.attnpool.q_proj.weight") == 3 and of("backbone.visual.layer4.0.conv1.weight") == 3
    assert of("backbone.visual.layer3.0.conv1.weight") == 2 and of("backbone.visual.layer2.0.conv1.weight") == 1
    assert of("backbone.visual.conv1.weight") == 0 and of("backbone.visual.layer1.0.conv1.weight") == 0
