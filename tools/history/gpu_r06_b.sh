#!/bin/bash
# round 6, call b: per-tensor gradient-cosine table, whole suite (timed, 32 oracle threads), default bench line, split-K emulation,
# kernel trace (bn_finalize after the hoisted M2 loads), five more RCCL-in-graph runs
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r06b; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
OMP_NUM_THREADS=32 timeout 600 python tools/grad_cos_table.py $O/grad_cos_r50_config1.json > $O/grad_cos.log 2>&1; echo "grad cos table rc=$?"; tail -28 $O/grad_cos.log
[ -f tests/golden/grad_cos_r50_config1.json ] || cp $O/grad_cos_r50_config1.json tests/golden/grad_cos_r50_config1.json
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 --durations=15 > $O/gpu_suite.log 2>&1; echo "suite rc=$? in $(( $(date +%s) - t0 )) s"
tail -24 $O/gpu_suite.log | cut -c1-200
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-600 $O/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06b/bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", {k: d["roofline"].get(k) for k in ("frac", "achieved", "traffic_ratio", "traffic_commit")})
print("module_path", {k: (v.get("ms_per_step") if isinstance(v, dict) else v) for k, v in d.get("module_path", {}).items() if k != "what"})
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "train_step_s_runs")})
PY
timeout 600 python tools/splitk_emulation.py --tsv $O/splitk_emulation.tsv > $O/splitk.log 2>&1; echo "splitk rc=$?"; grep SPLITK $O/splitk.log | cut -c1-330
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r06b -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-module-path > $O/prof.log 2>&1
echo "prof rc=$?"; grep -o '"ms_per_step": [0-9.]*' $O/prof.log | head -1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv && head -45 "$f" | cut -c1-220
find $O/prof -name '*kernel_trace.csv' -size +20M -delete; find $O/prof -name '*.db' -size +20M -delete
ok=0
for i in $(seq 5); do
  MASTER_PORT=$((29700 + i)) timeout 180 python tools/dist1_check.py graph 4 > $O/dist1_graph_$i.log 2>&1 && grep -q "^mode graph -> launch graph graph_error None" $O/dist1_graph_$i.log && ok=$((ok+1))
done
echo "dist1_check graph: $ok / 5 ok"
