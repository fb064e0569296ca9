#!/bin/bash
# round 3, call AE: the other BASELINE.json configurations at the final state (bench.py --steps 100 --warmup 10)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03ae
python -c "import __graft_entry__ as g; g.build()" || exit 1
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config']['workload'][:90])" >> $L.ab.log 2>&1; }
: > $L.ab.log
run r50_416
run r101_416 --spec r101
run r50_480_L22 --size 480
run r50_416_b16 --batch 16
run r50_416_b32 --batch 32
cat $L.ab.log
