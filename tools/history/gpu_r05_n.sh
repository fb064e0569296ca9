#!/bin/bash
# Round 5, call N: the reference-policy test (fp16 autocast + GradScaler beside the HIP path, float64 judge), alone
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r05n
( time timeout 160 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -s -p no:cacheprovider -k reference_fp16_policy ) 2>&1 | grep -v "Warning\|warn\|amdgpu.ids" | tail -12 | cut -c1-260 > $L.log; cat $L.log
