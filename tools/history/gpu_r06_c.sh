#!/bin/bash
# round 6, call c: the tests added since call b
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r06c; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 1500 python -m pytest tests/test_module_gpu.py tests/test_ref_loop_gpu.py tests/test_dist_gpu.py tests/test_p2p_gpu.py tests/test_bench_launch.py -m gpu -q -s --timeout 900 --durations=8 > $O/tests.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|FAILED|Error|forced \{" $O/tests.log | cut -c1-900 | tail -15
