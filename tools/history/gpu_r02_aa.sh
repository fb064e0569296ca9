#!/bin/bash
# round 2, call AA: inference path (folded BatchNorm, graph runner) + library-owned RCCL communicators; kernels touched by them
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02aa
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids"
timeout 400 python -m pytest tests/test_infer_gpu.py tests/test_comm_gpu.py -m gpu -q -s 2>&1 | grep -v "$F" | tail -40 | cut -c1-600 > $L.new.log
timeout 300 python -m pytest tests/test_hip_ops.py tests/test_engine_gpu.py -m gpu -q -k "pack or conv_gemm or adam or tiny_step or eval_forward" 2>&1 | grep -v "$F" | tail -8 | cut -c1-300 > $L.old.log
timeout 300 python tools/latency.py --iters 300 2>&1 | grep "LATENCY\|Error\|error" | cut -c1-3000 > $L.latency.log
echo "=== new"; cat $L.new.log
echo "=== old"; cat $L.old.log
echo "=== latency"; cat $L.latency.log
