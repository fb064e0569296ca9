#!/bin/bash
# Round 4, call J (last): what the driver runs at round end, on the final state - smoke, `pytest -m gpu`, the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r04j
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | grep -v "$F" | tail -5 | cut -c1-300 > $L.smoke.log; cat $L.smoke.log
( time timeout 1150 python -m pytest tests/ -x -q -m gpu --durations=8 ) 2>&1 | grep -v "$F" | tail -24 | cut -c1-200 > $L.gpu_suite.log; tail -20 $L.gpu_suite.log
( time timeout 900 python bench.py ) 2>$L.bench.err | tee $L.bench_n1.json | cut -c1-600; tail -4 $L.bench.err | cut -c1-200
