"""HIP-graph capture that a c10d (RCCL) process group in the same process cannot kill.

ProcessGroupNCCL runs a watchdog thread that polls `hipEventQuery` on the end event of every collective issued OUTSIDE a
capture until it has seen it complete (one pass every 100 ms).  A stream capture in the default "global" mode makes such a
query from ANY thread illegal while it lasts: the watchdog's HIP check fails and c10d aborts the process (round 5: the
driver's run of tests/test_dist_gpu.py died with SIGABRT in `WorkNCCL::finishedGPUExecutionInternal` - the eager first
step had queued the gradient all-reduces of two communicators, and the capture of the second step began before the
watchdog had retired them).  Two independent measures, both applied by `graph()`:

  * the capture runs in `thread_local` mode: only THIS thread's calls are checked against the capture, other threads
    (the watchdog, a DataLoader's pin-memory thread) may query events and synchronise as they like;
  * before it starts, the device is idle and the watchdog has had time to retire every work it still holds
    (`drain_c10d`: device synchronisation + a few of its polling periods) - so that even a runtime that ignored the mode
    would find nothing left to query.

Every capture of the package (trainer, drop-in module, inference runner) goes through here.
"""
import contextlib
import os
import time

import torch

# the watchdog sleeps kWatchdogThreadSleepMillis = 100 ms between passes over its work list
_WATCHDOG_PERIODS_S = float(os.environ.get("CRIS_CAPTURE_DRAIN_S", "0.35"))
CAPTURE_MODE = os.environ.get("CRIS_CAPTURE_MODE", "thread_local")


def c10d_device_group_alive():
    """True when this process holds a torch.distributed group with a device (RCCL) backend - the only case with a watchdog"""
    try:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return "nccl" in str(dist.get_backend()).lower()
    except Exception:                            # noqa: BLE001 - a group that is being torn down
        return False


def drain_c10d(device=None):
    """device idle + the c10d watchdog's work list empty (see the module docstring)"""
    torch.cuda.synchronize(device)
    if c10d_device_group_alive() and _WATCHDOG_PERIODS_S > 0:
        time.sleep(_WATCHDOG_PERIODS_S)
        torch.cuda.synchronize(device)


@contextlib.contextmanager
def graph(g, pool=None, device=None, drain=True):
    """`with capture.graph(g, pool=...)`: torch.cuda.graph in thread_local capture mode after drain_c10d()"""
    if drain:
        drain_c10d(device)
    kw = {} if pool is None else {"pool": pool}
    with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE, **kw):
        yield g
