#!/bin/bash
# round 2, call AE: record pipeline + JPEG cases incl. progressive / per-component scans on the GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02ae
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids"
timeout 300 python -m pytest tests/test_records_gpu.py tests/test_jpeg_gpu.py -m gpu -q 2>&1 | grep -v "$F" | tail -25 | cut -c1-500 > $L.tests.log
echo "=== tests"; cat $L.tests.log
