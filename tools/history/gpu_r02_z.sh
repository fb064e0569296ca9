#!/bin/bash
# round 2, call Z: Adam underneath backward with a capped grid (CRIS_ADAM_SIDE_BLOCKS sweep) against the single pass
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02z
timeout 300 python -m pytest tests/test_engine_gpu.py tests/test_hip_ops.py -m gpu -q -k "adam" 2>&1 | grep -v "Warning\|warn\|return float\|Consider using\|amdgpu.ids" | tail -15 | cut -c1-400 > $L.tests.log
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 200 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('launch'), d['config'].get('graph_error'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run overlap0 CRIS_ADAM_OVERLAP=0
run side32 CRIS_ADAM_SIDE_BLOCKS=32
run side64 CRIS_ADAM_SIDE_BLOCKS=64
run side128 CRIS_ADAM_SIDE_BLOCKS=128
run side256 CRIS_ADAM_SIDE_BLOCKS=256
run side512 CRIS_ADAM_SIDE_BLOCKS=512
run overlap0b CRIS_ADAM_OVERLAP=0
echo "=== tests"; cat $L.tests.log
echo "=== ab"; cat $L.ab.log; tail -3 $L.side128.err
