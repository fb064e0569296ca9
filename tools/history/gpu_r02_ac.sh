#!/bin/bash
# round 2, call AC: JPEG decode on the GPU (bit exactness against libjpeg-turbo, chain into the preprocessing kernel) + rates
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02ac
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids"
timeout 300 python -m pytest tests/test_jpeg_gpu.py tests/test_input_pipe.py -m gpu -q 2>&1 | grep -v "$F" | tail -25 | cut -c1-400 > $L.tests.log
timeout 200 python tools/jpeg_bench.py --iters 30 --threads 8 2>&1 | grep "JPEGBENCH\|Error" | cut -c1-1500 > $L.bench.log
timeout 200 python tools/jpeg_bench.py --iters 30 --threads 32 --batch 64 2>&1 | grep "JPEGBENCH\|Error" | cut -c1-1500 >> $L.bench.log
nproc >> $L.bench.log
echo "=== tests"; cat $L.tests.log
echo "=== bench"; cat $L.bench.log
