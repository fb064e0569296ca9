#!/bin/bash
# round 3, call M: 8-wave weight-gradient tile: tests, standalone timing on / off, step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03m
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_wgrad_tr_layout.py -m gpu -q -x -k "wgrad or dgrad" 2>&1 | grep -v "$F" | tail -12 | cut -c1-600 > $L.kernel_tests.log
echo "=== kernel tests"; cat $L.kernel_tests.log
CRIS_WGRAD8=1 timeout 200 python tools/wgrad_bench.py 2>&1 | grep "WGRAD8" > $L.wg1.log
CRIS_WGRAD8=0 timeout 200 python tools/wgrad_bench.py 2>&1 | grep "WGRAD8" > $L.wg0.log
echo "=== wgrad standalone"; paste -d'\n' $L.wg1.log $L.wg0.log
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 200 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run w8on CRIS_WGRAD8=1
run w8off CRIS_WGRAD8=0
run w8on_b512 CRIS_WGRAD8=1 CRIS_WGRAD8_BLOCKS=512
run w8on2 CRIS_WGRAD8=1
echo "=== step A/B"; cat $L.ab.log; tail -3 $L.w8on.err | cut -c1-300
