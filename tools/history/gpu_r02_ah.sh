#!/bin/bash
# round 2, call AH: kernel trace of the JPEG reconstruction kernels (rocprofv3 durations beside the HIP-event figure of call AC)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02ah
rm -rf gpurun_out/prof
timeout 40 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o jpeg -- python tools/jpeg_bench.py --iters 10 --batch 64 --threads 32 > $L.prof.log 2>&1
db=$(find gpurun_out/prof -name "*.db" | head -1); python tools/prof_summary.py $db $L.jpeg_kernel_stats.csv > $L.summary.log 2>&1
rm -rf gpurun_out/prof
cat $L.summary.log; head -5 $L.jpeg_kernel_stats.csv | cut -c1-200; grep JPEGBENCH $L.prof.log | cut -c1-700
