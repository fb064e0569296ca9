#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03o
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "wgrad" 2>&1 | grep -v "$F" | tail -5 | cut -c1-400 > $L.kernel_tests.log
echo "=== kernel tests"; cat $L.kernel_tests.log
CRIS_WGRAD8=1 CRIS_WGRAD8_MI=1 timeout 200 python tools/wgrad_bench.py 2>&1 | grep "WGRAD8" | sed 's/WGRAD8=1/MI=1/' > $L.wg1.log
CRIS_WGRAD8=1 CRIS_WGRAD8_MI=0 timeout 200 python tools/wgrad_bench.py 2>&1 | grep "WGRAD8" | sed 's/WGRAD8=1/MI=0/' > $L.wg0.log
CRIS_WGRAD8=1 CRIS_WGRAD8_MI=1 timeout 100 python tools/wgrad_repro.py 2>&1 | grep REPRO | cut -c1-200 > $L.repro.log
echo "=== wgrad standalone"; paste -d'\n' $L.wg1.log $L.wg0.log; cat $L.repro.log
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 200 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run w8off CRIS_WGRAD8=0
run w8big CRIS_WGRAD8=1 CRIS_WGRAD8_MIN_M=16384 CRIS_WGRAD8_MIN_K=4096
run w8all CRIS_WGRAD8=1
run w8off2 CRIS_WGRAD8=0
echo "=== step A/B"; cat $L.ab.log
