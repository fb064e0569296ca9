#!/bin/bash
# round 3, call AD: the module path at the final state, beside the native trainer on the same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03ad
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 300 python bench.py --path module --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer > $L.module.json 2> $L.module.err
echo "=== module"; cut -c1-330 $L.module.json; tail -1 $L.module.err | cut -c1-200
timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer 2>/dev/null | cut -c1-260
