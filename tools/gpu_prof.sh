#!/bin/bash
# rocprofv3 kernel trace of bench.py in HIP-graph mode -> gpurun_out/prof/<tag>_results.db
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; tag=${1:-r01}
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o $tag -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer > gpurun_out/prof.log 2>&1
echo "prof rc=$?"; grep '"metric"' gpurun_out/prof.log | cut -c1-260
