#!/bin/bash
# round 2, call J: eval post-processing kernels, checkpoint resume, DataParallel wrapper, trajectory bounds, dynconv geometry
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02j
timeout 600 python -m pytest tests/test_eval_post.py tests/test_module_gpu.py -q -m gpu 2>&1 | tail -12 > $L.a.log
timeout 600 python -m pytest tests/test_hip_ops.py -q 2>&1 | tail -4 > $L.b.log
timeout 900 python -m pytest tests/test_engine_gpu.py -q -s -k "trajectory or eval_forward or deterministic" 2>&1 | grep -v "Warning\|warn\|return float\|Consider using\|amdgpu.ids" | cut -c1-600 | tail -14 > $L.c.log
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timer 2>/dev/null | cut -c1-330 > $L.bench.log
for f in a b c bench; do echo "=== $f"; cat $L.$f.log; done
