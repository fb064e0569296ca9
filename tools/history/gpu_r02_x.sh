#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python tools/gemm_k_sweep.py 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r02x.log
