#!/bin/bash
# round 2, call K: the whole GPU test suite at HEAD + smoke + the bench line with the CPU baseline
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02k
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d" | cut -c1-500 > $L.gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $L.smoke.log 2>&1; echo "smoke rc=$?" >> $L.smoke.log
timeout 400 python bench.py --steps 200 --warmup 20 --shape-table $L.shapes.tsv > $L.bench.json 2> $L.bench.err
echo "=== gputests"; grep -n "passed\|failed\|FAILED\|bicubic max\|eval forward\|trajectory:\|max |hip" $L.gputests.log | cut -c1-300
echo "=== smoke"; tail -2 $L.smoke.log | cut -c1-300
echo "=== bench"; cut -c1-500 $L.bench.json; tail -2 $L.bench.err
