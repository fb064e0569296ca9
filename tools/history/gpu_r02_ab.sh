#!/bin/bash
# round 2, call AB: inference epilogue variant (EPI 2) - kernel test, inference + comm tests again, latency table
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02ab
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids"
timeout 400 python -m pytest tests/test_hip_ops.py tests/test_infer_gpu.py tests/test_comm_gpu.py -m gpu -q -s -k "folded or infer or graph_replay or refolds or rccl or conv_gemm" 2>&1 | grep -v "$F" | grep "folded vs\|passed\|failed\|FAILED\|Error\|assert\|COMM\|'rccl'" | cut -c1-900 > $L.tests.log
timeout 300 python tools/latency.py --iters 300 --modes fold+graph,nofold+graph 2>&1 | grep "LATENCY\|Error\|error" | cut -c1-3000 > $L.latency.log
echo "=== tests"; cat $L.tests.log
echo "=== latency"; cat $L.latency.log
