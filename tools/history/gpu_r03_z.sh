#!/bin/bash
# round 3, call Z: the default bench + kernel trace once more (call Y landed on a box whose MFMA-heavy kernels ran ~1.4x slower)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03z
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 400 python bench.py --shape-table $L.shapes.tsv > $L.bench.json 2> $L.bench.err
echo "=== bench"; cut -c1-400 $L.bench.json; tail -2 $L.bench.err | cut -c1-200
timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer 2>/dev/null | cut -c1-300
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r03z -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timer > $L.prof.log 2>&1
python tools/prof_summary.py $(ls gpurun_out/prof/*/r03z_results.db gpurun_out/prof/r03z_results.db 2>/dev/null | head -1) $L.kernel_stats.csv 40 "void adam_kernel<1>" 2>&1 | tail -3
rm -rf gpurun_out/prof
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|mclk" | head -8
