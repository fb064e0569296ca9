#!/bin/bash
# round 6, call r: final state after the arena exchange / wall-time mailbox limit: the driver's suite, smoke, default bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r06r; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 --durations=12 > $O/gpu_suite.log 2>&1; echo "suite rc=$? in $(( $(date +%s) - t0 )) s"
tail -18 $O/gpu_suite.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py --shape-table $O/gemm_shapes.tsv > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06r/bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", {k: d["roofline"].get(k) for k in ("frac", "achieved", "traffic_ratio", "traffic_commit")})
print("module_path", {k: (v.get("ms_per_step") if isinstance(v, dict) else v) for k, v in d.get("module_path", {}).items() if k != "what"})
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value", "cores")}, "commit", d["config"].get("source_commit"))
PY
