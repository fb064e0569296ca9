#!/bin/bash
# round 3, call F: module-path graphs (test + bench), in-step profile of the current build (kernel trace + shape table)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03f
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 600 python -m pytest tests/test_module_gpu.py tests/test_ref_loop_gpu.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -8 | cut -c1-500 > $L.tests.log
echo "=== module tests"; cat $L.tests.log
timeout 300 python bench.py --path module --steps 100 --warmup 5 > $L.module.json 2> $L.module.err
echo "=== module path (graphs)"; cut -c1-600 $L.module.json; tail -2 $L.module.err | cut -c1-300
CRIS_MODULE_GRAPH=0 timeout 300 python bench.py --path module --steps 30 --warmup 5 > $L.module_eager.json 2> $L.module_eager.err
echo "=== module path (eager)"; cut -c1-300 $L.module_eager.json
timeout 400 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --shape-table $L.shapes.tsv > $L.bench.json 2> $L.bench.err
echo "=== bench"; cut -c1-1500 $L.bench.json; tail -2 $L.bench.err | cut -c1-200
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r03f -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timer > $L.prof.log 2>&1
python tools/prof_summary.py $(ls gpurun_out/prof/*/r03f_results.db gpurun_out/prof/r03f_results.db 2>/dev/null | head -1) $L.kernel_stats.csv 45 2>&1 | tail -2
head -45 $L.kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/prof
