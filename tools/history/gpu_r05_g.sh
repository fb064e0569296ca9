#!/bin/bash
# Round 5, call G: what the driver runs at round end, on the final state - smoke, `pytest -m gpu`, the default bench line - and the
# kernel trace of that state
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r05g
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | grep -v "$F" | tail -5 | cut -c1-300 > $L.smoke.log; cat $L.smoke.log
( time timeout 1190 python -m pytest tests/ -x -q -m gpu --durations=10 -p no:cacheprovider ) 2>&1 | grep -v "$F" | tail -30 | cut -c1-220 > $L.gpu_suite.log; tail -24 $L.gpu_suite.log
( time timeout 900 python bench.py --shape-table $L.gemm_shapes.tsv ) > $L.bench_n1.json 2>$L.bench.err; echo "bench stdout lines: $(wc -l < $L.bench_n1.json)"; cut -c1-700 $L.bench_n1.json; tail -4 $L.bench.err | cut -c1-200
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r05 -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-module-path > $L.prof.log 2>&1
echo "prof rc=$?"; grep '"metric"' $L.prof.log | cut -c1-200
db=$(find gpurun_out/prof -name "*_results.db" | head -1)
python tools/prof_summary.py $db $L.kernel_stats.csv 40 "void adam_kernel<1>" | tail -6
rm -rf gpurun_out/prof
