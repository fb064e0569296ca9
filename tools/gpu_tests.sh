#!/bin/bash
# GPU test run with per-file logs (a GPU memory fault aborts the whole python process: keep the files separate)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for f in ${@:-tests/test_hip_ops.py tests/test_module_gpu.py tests/test_dist_gpu.py tests/test_engine_gpu.py}; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -m gpu -q -x -s --timeout 600 > gpurun_out/$n.log 2>&1
  echo "$n rc=$?"; grep -E "passed|failed|error|Memory access fault|trajectory" gpurun_out/$n.log | cut -c1-600 | tail -4
done
