#!/bin/bash
# Round 4, call A (parity closure):
#   1. the two code paths committed without a GPU run (CRIS_LN_BWD_V, CRIS_WGRAD_REDUCE_GROUP): kernel tests, engine tests, step A/B
#   2. the long parity runs (marker gpu_long): teacher-forced R50 x100, R101 x20, 480/L22 x20, stage-isolated at R50 full size
#   3. the per-stage / per-kind bf16 error budget (tools/error_budget.py) for R50 and R101
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r04a
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -v "^=\|^$" | head -30 > $L.smi_start.log
CRIS_TEST_NEXT=1 timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "narrow_instantiations or deferred_grouped or layernorm" 2>&1 | grep -v "$F" | tail -5 | cut -c1-600 > $L.kernel_tests.log
echo "=== kernel tests"; cat $L.kernel_tests.log
CRIS_LN_BWD_V=1 CRIS_WGRAD_REDUCE_GROUP=1 timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x --durations=25 2>&1 | grep -v "$F" | tail -40 | cut -c1-400 > $L.engine_tests.log
echo "=== engine tests (narrow LN backward + grouped reductions)"; tail -32 $L.engine_tests.log
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 200 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run base X=1
run lnv CRIS_LN_BWD_V=1
run lnv_1024 CRIS_LN_BWD_V=1 CRIS_LN_BWD_BLOCKS=1024
run redgroup CRIS_WGRAD_REDUCE_GROUP=1
run both CRIS_LN_BWD_V=1 CRIS_WGRAD_REDUCE_GROUP=1
run base2 X=1
echo "=== step A/B"; cat $L.ab.log
timeout 1500 python -m pytest tests/test_parity_long_gpu.py -m gpu_long -q -s --durations=10 2>&1 | grep -v "$F" | grep "teacher-forced\|passed\|failed\|Error\|assert\|bottleneck\|\[\|trajectory: \|max |hip\|s call\|worst" | cut -c1-500 > $L.long.log
echo "=== long parity tests"; tail -60 $L.long.log
timeout 600 python tools/error_budget.py --spec r50 --steps 100 --every 4 2>&1 | grep BUDGET | cut -c1-2000 > $L.budget_r50.log
echo "=== error budget r50"; grep "summary" -A 40 $L.budget_r50.log
timeout 400 python tools/error_budget.py --spec r101 --steps 20 --every 4 2>&1 | grep BUDGET | cut -c1-2000 > $L.budget_r101.log
echo "=== error budget r101"; grep "summary" -A 40 $L.budget_r101.log
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -v "^=\|^$" | head -30 > $L.smi_end.log
