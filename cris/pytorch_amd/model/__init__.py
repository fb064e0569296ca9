"""Drop-in for the reference's `model` package (reference model/__init__.py:1-49): `from model import build_segmenter`.

`build_segmenter(args)` returns `(model, param_list)` with the reference's two Adam parameter groups: group A = parameters
whose name starts with `backbone` and does not contain `positional_embedding` (`initial_lr = lr_multi * base_lr`), group B =
everything else, both backbone positional embeddings included (`initial_lr = base_lr`) - the rule of
model/__init__.py:36-48, which is why the HIP module keeps the reference's parameter names."""
from .segmenter import CRIS


def is_backbone_group(param_name: str) -> bool:
    return param_name.startswith("backbone") and "positional_embedding" not in param_name


def build_segmenter(args):
    net = CRIS(args)
    named = list(net.named_parameters())
    group_a = [p for name, p in named if is_backbone_group(name)]
    group_b = [p for name, p in named if not is_backbone_group(name)]
    groups = [dict(params=group_a, initial_lr=args.lr_multi * args.base_lr),
              dict(params=group_b, initial_lr=args.base_lr)]
    return net, groups


__all__ = ["CRIS", "build_segmenter", "is_backbone_group"]
