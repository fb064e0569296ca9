#!/bin/bash
# round 2, call L: weight gradients on the text-encoder stream (A/B in one box), eval post-processing re-test
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02l
timeout 600 python -m pytest tests/test_eval_post.py tests/test_hip_ops.py -q -m gpu 2>&1 | tail -4 > $L.a.log
CRIS_WGRAD_SIDE=1 timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_module_gpu.py -q -k "tiny_step or ragged or deterministic or stage_isolated or r50_small or launch_modes or config1 or other_shapes" 2>&1 | tail -6 > $L.b.log
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; env "$@" timeout 300 $B 2>$L.err_$tag | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), d['config']['final_loss'], d['config']['launch'], d['config']['graph_error'])" >> $L.ab.log 2>&1; }
: > $L.ab.log
run side0 CRIS_WGRAD_SIDE=0
run side1 CRIS_WGRAD_SIDE=1
run side0b CRIS_WGRAD_SIDE=0
run side1b CRIS_WGRAD_SIDE=1
run side1_r101 CRIS_WGRAD_SIDE=1 X=1
for f in a b ab; do echo "=== $f"; tail -8 $L.$f.log | cut -c1-300; done; tail -3 $L.err_side1
