#!/bin/bash
# round 2, call AF: runtime environment knobs against the default (launch latency of ~1040 dependent kernels per step)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02af
B="python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 120 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('launch'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run base A=1
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run noscratchreclaim HSA_NO_SCRATCH_RECLAIM=1
run hwq8 GPU_MAX_HW_QUEUES=8
run base2 A=1
echo "=== ab"; cat $L.ab.log
