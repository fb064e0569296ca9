#!/bin/bash
# round 6, call e (final state): the driver's suite (timed), smoke, the default bench line + shape table, kernel trace, HBM-traffic and SQ
# counter passes (separate --pmc runs, --kernel-trace only) -> gpurun_out/r06e/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r06e; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 --durations=12 > $O/gpu_suite.log 2>&1; echo "suite rc=$? in $(( $(date +%s) - t0 )) s"
tail -18 $O/gpu_suite.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; grep smoke: $O/smoke.log | cut -c1-400
timeout 900 python bench.py --shape-table $O/gemm_shapes.tsv > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06e/bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", {k: d["roofline"].get(k) for k in ("frac", "achieved", "traffic_ratio", "traffic_commit")})
print("module_path", {k: (v.get("ms_per_step") if isinstance(v, dict) else v) for k, v in d.get("module_path", {}).items() if k != "what"})
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "train_step_s_runs")})
PY
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r06e -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-module-path > $O/prof.log 2>&1
echo "prof rc=$?"; python tools/prof_summary.py $O/prof/r06e_results.db $O/kernel_stats.csv 40 "void adam_kernel<1>" | tail -2
rm -rf $O/prof
for c in FETCH_SIZE WRITE_SIZE; do
  d=$O/pmc_$c; rm -rf $d
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $d -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-module-path --launch eager > $O/pmc_$c.log 2>&1
  echo "$c rc=$?"
done
python tools/pmc_summary.py $O/pmc_FETCH_SIZE/pmc_results.db $O/pmc_WRITE_SIZE/pmc_results.db $O/hbm_traffic.json 5 | head -8
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
d=$O/pmc_sq; rm -rf $d
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $d -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-module-path --launch eager > $O/pmc_sq.log 2>&1
echo "sq rc=$?"; python tools/pmc_sq_summary.py $d/pmc_results.db $O/sq_counters.json | head -12; rm -rf $d
