#!/bin/bash
# round 2, call C: LDS attention, Adam-written packs, parallel wgrad reduction, reference-recipe test; bench + kernel trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02c
timeout 600 python -m pytest tests/test_hip_ops.py -q 2>&1 | tail -30 > $L.hip_ops.log
timeout 600 python -m pytest tests/test_engine_gpu.py -q -s -k "tiny_step or ragged or deterministic or eval_forward or stage_isolated or config1" 2>&1 | grep -v Warning | tail -30 > $L.engine.log
timeout 900 python -m pytest tests/test_ref_loop_gpu.py tests/test_module_gpu.py tests/test_dist_gpu.py -q -s 2>&1 | grep -v Warning | tail -30 > $L.loop.log
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --shape-table $L.shapes.tsv > $L.bench.json 2> $L.bench.err
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timer > $L.prof.log 2>&1
db=$(find gpurun_out/prof -name "*.db" | head -1); python tools/prof_summary.py $db $L.kernel_stats.csv 44 > $L.prof_summary.log 2>&1
rm -rf gpurun_out/prof
timeout 300 python - > $L.traj.log 2>&1 <<'PY'
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_engine_gpu as T
losses, ref, diffs = T._trajectory("traj_r50_b8_s416_d0.1_lr0.0001.json")
print("steps", len(losses), "max %.4e mean %.4e" % (max(diffs), sum(diffs) / len(diffs)))
print(" ".join("%.4f/%.4f" % (a, b) for a, b in zip(losses, ref)))
PY
for f in hip_ops engine loop traj prof_summary; do echo "=== $f"; tail -22 $L.$f.log; done
echo "=== bench"; cut -c1-700 $L.bench.json; tail -3 $L.bench.err; head -30 $L.kernel_stats.csv | cut -c1-150
