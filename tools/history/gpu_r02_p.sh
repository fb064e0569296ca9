#!/bin/bash
# round 2, call P: input pipeline parity + throughput; determinism tests; trajectory dump
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02p.log; : > $L
timeout 600 python -m pytest tests/test_input_pipe.py tests/test_abi.py -q -x 2>&1 | tail -6 >> $L
timeout 300 python tools/input_pipe_bench.py 200 2>&1 | grep -v "amdgpu.ids" >> $L
timeout 900 python -m pytest tests/test_engine_gpu.py -q -x -s -k "two_streams or deterministic or trajectory_r50" 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider\|return float" | cut -c1-300 | tail -12 >> $L
cat $L
