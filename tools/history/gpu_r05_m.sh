#!/bin/bash
# Round 5, call M (last): `pytest -m gpu` on the final commit
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r05m
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
( time timeout 1000 python -m pytest tests/ -x -q -m gpu --durations=6 -p no:cacheprovider ) 2>&1 | grep -v "$F" | tail -26 | cut -c1-220 > $L.gpu_suite.log; tail -22 $L.gpu_suite.log
