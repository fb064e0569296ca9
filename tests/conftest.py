import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun / the driver's GPU tier)")
    config.addinivalue_line("markers", "gpu_long: GPU parity runs of a minute or more each (100-step trajectories, teacher-forced "
                                       "runs of every BASELINE configuration); select with -m 'gpu or gpu_long'")


# GPU run order: deterministic per-kernel parity first, then the whole-network comparisons, then the statistical ones
# (loss trajectories, multi-process runs) - under `-x` a failure in a late, noise-sensitive test must not hide the kernel tests.
_GPU_ORDER = ["test_hip_ops", "test_engine_gpu", "test_parity_long_gpu", "test_module_gpu", "test_dist_gpu", "test_ref_loop_gpu", "test_eval_post", "test_input_pipe", "test_jpeg_gpu", "test_records_gpu", "test_p2p_gpu", "test_comm_gpu"]


def _gpu_rank(item):
    name = os.path.basename(str(item.fspath))
    for i, stem in enumerate(_GPU_ORDER):
        if name.startswith(stem):
            return 9 if "trajectory" in item.name else i     # the two loss-trajectory runs go last of all
    return -1                                    # CPU-only files keep their place in front


def pytest_collection_modifyitems(config, items):
    import torch
    items.sort(key=_gpu_rank)                    # stable sort: the order inside a file is unchanged
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords or "gpu_long" in it.keywords:
            it.add_marker(skip)
