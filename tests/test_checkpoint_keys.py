"""Checkpoint key interchange with the reference: the reference saves `model.state_dict()` of its
DistributedDataParallel-wrapped model (train.py:192-204: every key starts with `module.`) and test.py:74-78 loads that file
strictly into `torch.nn.DataParallel(model)`.  The native trainer must read such a file and be able to write one."""
import os

import torch

from cris.pytorch_amd import arch
from cris.pytorch_amd.trainer import NativeTrainer, strip_ddp_prefix

from conftest import GOLDEN


def _trainer(seed):
    clip, head = arch.specs_by_name("tiny")
    return clip, head, NativeTrainer(clip, head, arch.synthetic_state_dict(clip, head, seed), "cpu")


def test_reference_style_checkpoint_round_trip():
    clip, head, tr = _trainer(0)
    sd = tr.model_state_dict(ddp_prefix=True)
    assert all(k.startswith("module.") for k in sd)
    # the reference's test.py loader: strict load into a DataParallel wrapper of the module (same parameter tree / key list)
    wrapped = torch.nn.DataParallel(arch.build_param_tree(clip, head))
    assert list(sd.keys()) == list(wrapped.state_dict().keys())            # same keys, same order as the reference writes
    res = wrapped.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    # and back: a checkpoint written by the reference (DDP keys) loads into a trainer built from other weights
    _, _, other = _trainer(1)
    ref_ckpt = {"epoch": 3, "state_dict": wrapped.state_dict()}
    other.load_model_state_dict(ref_ckpt["state_dict"])
    a, b = other.model_state_dict(), tr.model_state_dict()
    assert list(a) == list(b)
    assert all(torch.equal(a[k], b[k]) for k in a if not k.endswith("num_batches_tracked"))
    # the constructor takes the prefixed form too
    third = NativeTrainer(clip, head, sd, "cpu")
    c = third.model_state_dict()
    assert all(torch.equal(c[k], b[k]) for k in b if not k.endswith("num_batches_tracked"))


def test_key_list_is_the_reference_list():
    clip, head = arch.specs_by_name("r50")
    want = [l.split()[0] for l in open(os.path.join(GOLDEN, "state_dict_keys_r50.txt")) if l.strip()]
    tree = arch.build_param_tree(clip, head)
    assert list(tree.state_dict().keys()) == want
    assert list(strip_ddp_prefix({"module." + k: 0 for k in want}).keys()) == want
    mixed = {"module.a": 1, "b": 2}
    assert strip_ddp_prefix(mixed) is mixed                                # only a uniformly prefixed dict is rewritten


def test_missing_key_is_reported():
    _, _, tr = _trainer(0)
    sd = tr.model_state_dict()
    sd.pop(next(iter(sd)))
    try:
        tr.load_model_state_dict(sd)
    except KeyError as e:
        assert "lacks 1 keys" in str(e)
    else:
        raise AssertionError("a truncated state_dict was accepted")
