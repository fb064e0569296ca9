#!/bin/bash
# round 2, call O: whole GPU suite + smoke + bench + kernel trace + shape table at the no-packed-fp32 build
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02o
timeout 1700 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d" | cut -c1-500 > $L.gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $L.smoke.log 2>&1; echo "smoke rc=$?" >> $L.smoke.log
timeout 400 python bench.py --steps 200 --warmup 20 --shape-table $L.shapes.tsv > $L.bench.json 2> $L.bench.err
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timer > $L.prof.log 2>&1
db=$(find gpurun_out/prof -name "*.db" | head -1); python tools/prof_summary.py $db $L.kernel_stats.csv 44 > $L.prof_summary.log 2>&1
rm -rf gpurun_out/prof
echo "=== gputests"; grep -n "passed\|failed\|FAILED\|Error" $L.gputests.log | cut -c1-300
echo "=== smoke"; tail -2 $L.smoke.log | cut -c1-300
echo "=== prof"; head -30 $L.prof_summary.log | cut -c1-200
echo "=== bench"; cut -c1-700 $L.bench.json; tail -2 $L.bench.err
