import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# The CPU oracle is the checker of most GPU tests, and on the GPU box's 128 cores torch's default (one thread per core) runs it
# 3.5x SLOWER than 32 threads do (bench.py cpu_baseline, round 5: 11.7 s against 3.3 s per R50 train step) - a third of the GPU
# suite's wall time was oversubscribed CPU work.  Children (spawned ranks, subprocess tests) inherit the environment variable.
if (os.cpu_count() or 1) > 32:
    os.environ.setdefault("OMP_NUM_THREADS", "32")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun / the driver's GPU tier)")
    config.addinivalue_line("markers", "gpu_long: GPU parity runs of a minute or more each (100-step trajectories, teacher-forced "
                                       "runs of every BASELINE configuration); select with -m 'gpu or gpu_long'")


# GPU run order (round-5 review: one aborting multi-process test in the middle hid every row behind it under `-x`): the
# deterministic single-process files first - kernels, whole network, drop-in module, inference, evaluation post-processing,
# input pipeline, long parity runs - and EVERY file that spawns processes or builds a process group last, the RCCL one last
# of all.  A failure in a late, noise-sensitive test cannot hide the kernel tests or the f1 / f2 rows.
_GPU_ORDER = ["test_hip_ops", "test_engine_gpu", "test_oracle_device", "test_module_gpu", "test_infer_gpu", "test_eval_post", "test_input_pipe",
              "test_jpeg_gpu", "test_png", "test_records_gpu", "test_parity_long_gpu",
              # multi-process from here on
              "test_bench_launch", "test_p2p_gpu", "test_comm_gpu", "test_ref_loop_gpu", "test_dist_gpu"]
_LAST = len(_GPU_ORDER) + 1


def _gpu_rank(item):
    name = os.path.basename(str(item.fspath))
    if "gpu" not in item.keywords and "gpu_long" not in item.keywords:
        return -1                                # CPU tests keep their place in front
    for i, stem in enumerate(_GPU_ORDER):
        if name.startswith(stem):
            if "trajectory" in item.name:
                return len(_GPU_ORDER) - 5.5     # the loss-trajectory runs: last of the single-process part
            if "rccl" in item.name:
                return _LAST                     # the one test that needs a c10d RCCL group: last of all
            return i
    return len(_GPU_ORDER) - 5.7                 # a GPU file this list does not know: before the multi-process part


def pytest_collection_modifyitems(config, items):
    import torch
    if (os.cpu_count() or 1) > 32 and torch.get_num_threads() > 32:
        torch.set_num_threads(32)
    items.sort(key=_gpu_rank)                    # stable sort: the order inside a file is unchanged
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords or "gpu_long" in it.keywords:
            it.add_marker(skip)
