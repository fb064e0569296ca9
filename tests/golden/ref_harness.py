"""Import the *reference* implementation (read-only at /root/reference) in THIS container.

Test infrastructure only.  Used by tests/golden/make_golden.py to pin oracle/ against the
reference's own forward/loss/grad, and by nothing that runs on the GPU box (the reference
does not exist there).  Recipe follows SURVEY.md section 8(c): stub the missing third-party
modules the reference imports (loguru, wandb, cv2, lmdb, ftfy), put /root/reference on
sys.path, and replace torch.jit.load (the CLIP archive pretrain/RN50.pt is not shipped)
with an object that hands back a caller-provided CLIP state_dict.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("CRIS_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REF_ROOT, "model", "segmenter.py"))


def _install_stubs():
    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    if "loguru" not in sys.modules:
        m = types.ModuleType("loguru")
        m.logger = _Logger()
        sys.modules["loguru"] = m
    if "ftfy" not in sys.modules:
        m = types.ModuleType("ftfy")
        m.fix_text = lambda s: s
        sys.modules["ftfy"] = m
    for name in ("cv2", "lmdb", "wandb"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)


def import_reference():
    """Returns the reference `model` package (model.segmenter, model.clip, model.layers)."""
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REF_ROOT)
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import model  # noqa: the reference's package
    import model.clip
    import model.layers
    import model.segmenter
    return model


class _FakeJit:
    def __init__(self, sd):
        self._sd = sd

    def eval(self):
        return self

    def state_dict(self):
        return dict(self._sd)


def build_reference_cris(clip_state_dict, cfg_dict):
    """model.segmenter.CRIS(cfg) with torch.jit.load patched to yield clip_state_dict."""
    import torch
    ref = import_reference()
    from utils.config import CfgNode
    orig = torch.jit.load
    torch.jit.load = lambda *a, **k: _FakeJit(clip_state_dict)
    try:
        net = ref.segmenter.CRIS(CfgNode(dict(cfg_dict)))
    finally:
        torch.jit.load = orig
    return net
