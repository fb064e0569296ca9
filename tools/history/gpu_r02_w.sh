#!/bin/bash
# round 2, call W: whole GPU suite + smoke + default bench line at HEAD; the other BASELINE configurations
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02w
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -v "UserWarning\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d\|out.append" | tail -12 | cut -c1-300 > $L.gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $L.smoke.log 2>&1; echo "smoke rc=$?" >> $L.smoke.log
timeout 600 python bench.py --shape-table $L.shapes.tsv > $L.bench.json 2> $L.bench.err
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 300 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config']['workload'][:90])" >> $L.ab.log 2>&1; }
: > $L.ab.log
run r50_416
run r101_416 --spec r101
run r50_480_L22 --size 480
run r50_416_b16 --batch 16
run r50_416_b32 --batch 32
echo "=== gputests"; cat $L.gputests.log
echo "=== smoke"; tail -2 $L.smoke.log | cut -c1-300
echo "=== bench"; cut -c1-600 $L.bench.json; tail -2 $L.bench.err
echo "=== configs"; cat $L.ab.log
