#!/bin/bash
# round 2, call Y: Adam underneath the backward pass (trainer.adam_overlap): bit-identity test + A/B of the step time
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02y
timeout 300 python -m pytest tests/test_engine_gpu.py -m gpu -q -k "adam_under_backward or training_step_is_deterministic" 2>&1 | grep -v "Warning\|warn\|return float\|Consider using\|amdgpu.ids" | tail -15 | cut -c1-400 > $L.tests.log
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 200 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('launch'), d['config'].get('graph_error'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run overlap1 CRIS_ADAM_OVERLAP=1
run overlap0 CRIS_ADAM_OVERLAP=0
run overlap1b CRIS_ADAM_OVERLAP=1
echo "=== tests"; cat $L.tests.log
echo "=== ab"; cat $L.ab.log; tail -3 $L.overlap1.err
