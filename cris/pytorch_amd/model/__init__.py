"""Drop-in for the reference's `model` package (reference model/__init__.py:1-49): `from model import build_segmenter`.

`build_segmenter(args)` returns `(model, param_list)` with the reference's two Adam parameter groups - names starting with
`backbone` but not containing `positional_embedding` get `initial_lr = lr_multi * base_lr`, everything else (including
both backbone positional embeddings) `initial_lr = base_lr` (model/__init__.py:36-48)."""
from .segmenter import CRIS


def build_segmenter(args):
    model = CRIS(args)
    backbone, head = [], []
    for k, v in model.named_parameters():
        if k.startswith("backbone") and "positional_embedding" not in k:
            backbone.append(v)
        else:
            head.append(v)
    param_list = [{"params": backbone, "initial_lr": args.lr_multi * args.base_lr},
                  {"params": head, "initial_lr": args.base_lr}]
    return model, param_list


__all__ = ["CRIS", "build_segmenter"]
