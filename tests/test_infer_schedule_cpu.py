"""The inference schedule without a GPU: cris.pytorch_amd.infer.InferEngine run with every library launch replaced by a
recorder (no arithmetic happens - this is host logic only): which BatchNorms are folded, what the fused GEMM launches carry
(bias = the BatchNorm shift, ReLU before / after the residual), how many launches an eval forward is.  The numerics of the
same schedule are tested on the GPU (tests/test_infer_gpu.py)."""
import collections
import dataclasses

import pytest
import torch

from cris.pytorch_amd import arch, hip, ops, synth
from cris.pytorch_amd.infer import InferenceRunner


class _Stream:
    cuda_stream = 0

    def wait_stream(self, other):
        pass


@pytest.fixture
def recorder(monkeypatch):
    log = []
    monkeypatch.setattr(hip, "call", lambda name, *args: log.append((name, args)))
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _Stream())
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    assert ops.hip is hip
    return log


@pytest.mark.parametrize("spec,batch,size,word_len", [("tiny", 3, 96, 9), ("r50", 1, 416, 17), ("r101", 2, 480, 22)])
def test_folded_schedule(recorder, spec, batch, size, word_len):
    clip, head = arch.specs_by_name(spec)
    head = dataclasses.replace(head, word_len=word_len)
    sd = arch.synthetic_state_dict(clip, head, 0)
    img, word, _ = synth.make_batch(batch, size, word_len, 0, 0)
    counts = {}
    for fold in (True, False):
        r = InferenceRunner(clip, head, sd, torch.device("cpu"), fold_bn=fold, use_graph=False)
        r(img, word)                       # first call: packs and folds
        del recorder[:]
        out = r(img, word)
        assert tuple(out.shape) == (batch, 1, size // 4, size // 4)
        names = collections.Counter(n for n, _ in recorder)
        counts[fold] = names
        e = r.engine
        n_bn = len(e.bn_prefixes)
        if not fold:
            assert len(e._fold) == 0 and names["cris_bn_eval_coeffs"] == n_bn
            continue
        # not folded: the two apply kernels around the per-sample multiplication (neck.f1_v_proj / norm_layer) - and, with
        # CRIS_STATE_FP32=1, the BatchNorm1d of neck.txt_proj, which then runs in fp32 on the sentence vector's rows (csrc/smallf32.hip)
        extra = 0
        if e.state_f32:
            assert names["cris_bn_relu_f32_small"] == 1 and names["cris_linear_f32_small"] == 3 and names["cris_eot_gather_ln_f32"] == 1
            n_bn, extra = n_bn - 1, 1
        assert len(e._fold) == n_bn - 2 and names["cris_bn_eval_coeffs"] == 2 + extra and names["cris_bn_apply"] == 2
        assert names["cris_pack_weights"] == 0                     # frozen weights: nothing is packed again
        gemms = [a[0]._obj for n, a in recorder if n == "cris_conv_gemm_variant"]
        shifts = {t[1].data_ptr() for t in e._fold.values()}
        fused = [p for p in gemms if p.bias in shifts]
        assert len(fused) == n_bn - 2                              # one fused launch per folded BatchNorm
        n_blocks = sum(clip.vision_layers)
        after = [p for p in fused if p.act == 3]
        assert len(after) == n_blocks + 1 and all(p.resid for p in after)          # Bottleneck tails + attnpool.connect
        assert all(p.act in (0, 1, 3) and not p.colsum for p in fused)
        assert sum(1 for p in fused if p.act == 0) == 4             # the downsample branch of each layer group's first block (model/clip.py:32-42)
    assert sum(counts[True].values()) < sum(counts[False].values()) - (len(e.bn_prefixes) - 2)

