#!/bin/bash
# Round 4, call N: what an fp32 text encoder (and stem) would buy - the bf16-storage emulation with every point on EXCEPT one
# stage's, along the oracle's own trajectory (configs[1], 25 states), next to the HIP path from the same states.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 135 python tools/error_budget.py --steps 100 --every 4 --without text+stem --out gpurun_out/error_budget_r50_without.json 2>&1 | grep BUDGET | cut -c1-400 > gpurun_out/r04n.budget_without.log
tail -12 gpurun_out/r04n.budget_without.log
