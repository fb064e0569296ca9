#!/bin/bash
# round 2, call M: is the full-size step bit-reproducible?  two trainers x 10 steps, graph and eager, then bisect by switch
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02m.log; : > $L
run() { echo "##### $*" >> $L; env "$@" timeout 300 python tools/determinism_check.py r50 8 416 8 ${MODE:-graph} 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | cut -c1-700 >> $L; }
MODE=graph run X=1
MODE=eager run X=1
MODE=eager run CRIS_ATTN_LDS_MIN=100000
MODE=eager run CRIS_ZERO_ALL=1
MODE=eager run CRIS_WGRAD_GROUP_M=0
echo "##### tiny graph" >> $L; timeout 200 python tools/determinism_check.py tiny 4 64 8 graph 2>&1 | grep -v "amdgpu.ids\|Warn" | cut -c1-400 >> $L
cat $L
