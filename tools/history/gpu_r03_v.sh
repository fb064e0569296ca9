#!/bin/bash
# round 3, call V: eight-deep-ring 64x64 tile for the M <= 144 problems (text encoder, K <= 512)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03v
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "8wave or skinny or adam or pack" 2>&1 | grep -v "$F" | tail -5 | cut -c1-400 > $L.kernel_tests.log
echo "=== kernel tests"; cat $L.kernel_tests.log
timeout 300 python tools/skinny_bench.py 2>&1 | grep "^SKINNY" > $L.skinny.log; cat $L.skinny.log
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 200 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run base X=1
run nodeep CRIS_GEMM_DEEP=0
run base2 X=1
run nodeep2 CRIS_GEMM_DEEP=0
echo "=== step A/B"; cat $L.ab.log
