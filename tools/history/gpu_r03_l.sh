#!/bin/bash
# round 3, call L: the whole GPU suite + smoke + bench + rocprofv3 kernel trace + PMC passes (HBM traffic, SQ counters)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03l
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=12 2>&1 | grep -v "$F" | tail -32 | cut -c1-300 > $L.gputests.log
echo "suite seconds: $(( $(date +%s) - t0 ))" >> $L.gputests.log
echo "=== gputests"; cat $L.gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $L.smoke.log 2>&1; echo "smoke rc=$?" >> $L.smoke.log; tail -2 $L.smoke.log | cut -c1-400
timeout 400 python bench.py --shape-table $L.shapes.tsv > $L.bench.json 2> $L.bench.err
echo "=== bench"; cut -c1-700 $L.bench.json; tail -2 $L.bench.err | cut -c1-200
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r03l -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timer > $L.prof.log 2>&1
python tools/prof_summary.py $(ls gpurun_out/prof/*/r03l_results.db gpurun_out/prof/r03l_results.db 2>/dev/null | head -1) $L.kernel_stats.csv 45 2>&1 | tail -2
rm -rf gpurun_out/prof
bash tools/gpu_pmc.sh > $L.pmc.log 2>&1
python tools/pmc_summary.py $(ls gpurun_out/pmc_FETCH_SIZE/*/pmc_results.db gpurun_out/pmc_FETCH_SIZE/pmc_results.db 2>/dev/null | head -1) $(ls gpurun_out/pmc_WRITE_SIZE/*/pmc_results.db gpurun_out/pmc_WRITE_SIZE/pmc_results.db 2>/dev/null | head -1) $L.hbm_traffic.json 2>&1 | tail -14
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
bash tools/gpu_pmc_sq.sh > $L.pmc_sq.log 2>&1; cp gpurun_out/pmc_sq_summary.json $L.sq_counters.json 2>/dev/null; tail -3 $L.pmc_sq.log | cut -c1-300
