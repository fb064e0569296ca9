#!/bin/bash
# round 2, call AG: fewer hardware queues (GPU_MAX_HW_QUEUES 8 doubled the step time in call AF; default 4)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02ag
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 100 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('launch'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run hwq2 GPU_MAX_HW_QUEUES=2
run hwq1 GPU_MAX_HW_QUEUES=1
run hwq3 GPU_MAX_HW_QUEUES=3
echo "=== ab"; cat $L.ab.log
