#!/bin/bash
# Round 5, call H: after the last (range-guard) edits of the reciprocal index arithmetic: kernel + engine tests; then the `gpu_long`
# set (free-running 100-step curve, the trainer with EIGHT ranks on one GPU, command-list replay with two ranks)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r05h
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
T() { tag=$1; shift; ( time timeout 1500 python -m pytest "$@" -q -x -p no:cacheprovider --durations=5 ) 2>&1 | grep -v "$F" | tail -16 | cut -c1-300 > $L.$tag.log; echo "=== $tag"; tail -10 $L.$tag.log; }
T kernels tests/test_hip_ops.py -m gpu
T engine tests/test_engine_gpu.py -m gpu -k "tiny or config1 or other_shapes or deterministic or r101 or long_text"
T long tests/ -m gpu_long
