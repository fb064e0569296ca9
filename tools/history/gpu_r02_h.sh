#!/bin/bash
# round 2, call H: restored BatchNorm finalize / apply geometry; the other BASELINE configurations
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02h
timeout 600 python -m pytest tests/test_hip_ops.py -q 2>&1 | tail -4 > $L.hip_ops.log
timeout 600 python -m pytest tests/test_engine_gpu.py -q -k "tiny_step or ragged or deterministic or r101_step or stage_isolated" 2>&1 | tail -6 > $L.engine.log
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 300 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config']['workload'][:90])" >> $L.ab.log 2>&1; }
: > $L.ab.log
run r50_416
run r50_416_again
run r101_416 --spec r101
run r50_480_L22 --size 480
run r50_416_b16 --batch 16
run r50_416_b32 --batch 32
for f in hip_ops engine ab; do echo "=== $f"; tail -10 $L.$f.log | cut -c1-300; done
