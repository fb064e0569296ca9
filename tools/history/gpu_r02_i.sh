#!/bin/bash
# round 2, call I: the whole GPU test suite at HEAD, then the committed bench line / kernel trace / shape table
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02i
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "Warning\|warn\|return float\|Consider using\|amdgpu.ids" | tail -15 > $L.gputests.log
timeout 400 python bench.py --steps 200 --warmup 20 --shape-table $L.shapes.tsv > $L.bench.json 2> $L.bench.err
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timer > $L.prof.log 2>&1
db=$(find gpurun_out/prof -name "*.db" | head -1); python tools/prof_summary.py $db $L.kernel_stats.csv 44 > $L.prof_summary.log 2>&1
rm -rf gpurun_out/prof
python -c "import __graft_entry__ as g; g.smoke()" > $L.smoke.log 2>&1; echo "smoke rc=$?" >> $L.smoke.log
echo "=== gputests"; cat $L.gputests.log | cut -c1-300
echo "=== smoke"; tail -3 $L.smoke.log | cut -c1-400
cat $L.prof_summary.log
echo "=== bench"; cut -c1-600 $L.bench.json; tail -2 $L.bench.err
