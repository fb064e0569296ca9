#!/bin/bash
# FIRST CALL OF THE NEXT ROUND: code written after round 3's GPU budget was spent (cross-compiled, default off, never run).
#   1. LayerNorm backward with 1 / 2 channel vectors per lane for C <= 512 / 1024 (CRIS_LN_BWD_V=1): 112 / 140 instead of 238 VGPRs
#      -> tests, then step A/B with the default and a larger grid (CRIS_LN_BWD_BLOCKS)
#   2. split reductions of the large weight gradients deferred to the queue flush and grouped (CRIS_WGRAD_REDUCE_GROUP=1): 32 -> ~6 launches
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/next_a
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
CRIS_TEST_NEXT=1 timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "layernorm or deferred_grouped" 2>&1 | grep -v "$F" | tail -5 | cut -c1-400 > $L.kernel_tests.log
echo "=== kernel tests"; cat $L.kernel_tests.log
CRIS_LN_BWD_V=1 CRIS_WGRAD_REDUCE_GROUP=1 timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "not 100_steps" 2>&1 | grep -v "$F" | tail -4 | cut -c1-400 > $L.engine_tests.log
echo "=== engine tests (narrow LN backward + grouped reductions)"; cat $L.engine_tests.log
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 200 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run base X=1
run lnv CRIS_LN_BWD_V=1
run lnv_1024 CRIS_LN_BWD_V=1 CRIS_LN_BWD_BLOCKS=1024
run lnv_1536 CRIS_LN_BWD_V=1 CRIS_LN_BWD_BLOCKS=1536
run redgroup CRIS_WGRAD_REDUCE_GROUP=1
run base2 X=1
echo "=== step A/B"; cat $L.ab.log
