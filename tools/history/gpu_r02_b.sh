#!/bin/bash
# round 2, call B: per-kernel parity of the deterministic kernels / grouped wgrad, whole-step parity, determinism, step time
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; L=gpurun_out/r02b
timeout 600 python -m pytest tests/test_hip_ops.py -q 2>&1 | tail -40 > $L.hip_ops.log
timeout 600 python -m pytest tests/test_engine_gpu.py -q -s -k "tiny_step or ragged or deterministic or eval_forward or stage_isolated" 2>&1 | grep -v Warning | tail -40 > $L.engine_small.log
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --shape-table $L.shapes.tsv > $L.bench.json 2> $L.bench.err
timeout 900 python -m pytest tests/test_engine_gpu.py -q -s -k "config1 or config3 or config4 or r50_small" 2>&1 | grep -v Warning | tail -30 > $L.engine_full.log
timeout 300 python - > $L.traj.log 2>&1 <<'PY'
import json, os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_engine_gpu as T
for fx in ("traj_r50_b8_s416_d0.1_lr0.0001.json", "traj_tiny_b4_s64_d0.1_lr0.0001.json"):
    losses, ref, diffs = T._trajectory(fx)
    print(fx, "steps", len(losses), "max %.4e mean %.4e" % (max(diffs), sum(diffs) / len(diffs)))
    print(" ".join("%.4f/%.4f" % (a, b) for a, b in zip(losses, ref)))
PY
timeout 600 python -m pytest tests/test_module_gpu.py tests/test_dist_gpu.py -q 2>&1 | tail -15 > $L.module_dist.log
for f in hip_ops engine_small engine_full traj module_dist; do echo "=== $f"; tail -25 $L.$f.log; done
echo "=== bench"; cut -c1-1500 $L.bench.json; tail -3 $L.bench.err
