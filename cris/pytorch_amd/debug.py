"""The ONE place for diagnostic switches of the training path.  Nothing here changes results; every switch defaults to off
and is read from the environment once, when the package is imported (tests and tools may also set the attributes of `HOOKS`
directly before they build an Engine / NativeTrainer).

| attribute            | environment          | effect                                                                                   | used by |
|----------------------|----------------------|------------------------------------------------------------------------------------------|---------|
| zero_all             | CRIS_ZERO_ALL=1      | clear the WHOLE gradient arena every step instead of only the accumulated ranges          | tests/test_engine_gpu.py (proves the other ranges are fully overwritten) |
| no_side_stream       | CRIS_NO_SIDE=1       | text encoder on the launch stream instead of its own                                      | tools/determinism_debug.py |
| force_dist           | CRIS_FORCE_DIST=1    | take the multi-rank code paths with a 1-rank communicator                                 | tools/dist1_check.py, tools/comm1_check.py |
| hold_backward_fork   | CRIS_DEBUG=sleep     | hold the device back at the backward fork so that both encoders' backward run concurrently | tests/test_engine_gpu.py, tools/determinism_check.py |
| taps                 | (set to a list)      | collect stream-ordered copies of the gradient buffers after every text-backward closure   | tools/determinism_check.py |
"""
import os


class _Hooks:
    def __init__(self):
        env = os.environ.get
        self.zero_all = env("CRIS_ZERO_ALL", "0") == "1"
        self.no_side_stream = env("CRIS_NO_SIDE", "0") == "1"
        self.force_dist = env("CRIS_FORCE_DIST", "0") == "1"
        self.hold_backward_fork = "sleep" in env("CRIS_DEBUG", "")
        self.taps = None


HOOKS = _Hooks()


def tap(i, fn):
    """copies (in stream order) of every gradient buffer the backward closure `fn` that just ran can see -> HOOKS.taps"""
    import torch
    from .engine import Act
    objs = list(fn.__defaults__ or ()) + [c.cell_contents for c in (fn.__closure__ or ())]
    rec = []
    for k, o in enumerate(objs):
        if isinstance(o, Act) and o.g is not None:
            rec.append(("%d.g" % k, o.g.clone()))
        elif isinstance(o, dict) and "buf" in o:
            b = o["buf"]
            for j, t in enumerate(b if isinstance(b, (tuple, list)) else (b,)):
                if torch.is_tensor(t):
                    rec.append(("%d.buf%d" % (k, j), t.clone()))
    extra = {}
    if fn.__qualname__.startswith("Engine.ln."):                     # the closure's own inputs, by cell name
        for nm, c in zip(fn.__code__.co_freevars, fn.__closure__ or ()):
            o = c.cell_contents
            if torch.is_tensor(o):
                extra[nm] = o.clone()
            elif isinstance(o, Act):
                extra[nm + ".t"] = o.t.clone()
                if o.g is not None:
                    extra[nm + ".g"] = o.g.clone()
    HOOKS.taps.append((i, fn.__qualname__, rec, extra))
