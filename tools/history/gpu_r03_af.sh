#!/bin/bash
# round 3, call AF: 64x64 tile with the K-steps split over two wave groups (CRIS_GEMM_KS2)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03af
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 300 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "64x64k2" 2>&1 | grep -v "$F" | tail -6 | cut -c1-300 > $L.kernel_tests.log
echo "=== kernel tests"; cat $L.kernel_tests.log
B="python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 150 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run base X=1
run ks2 CRIS_GEMM_KS2=1
run ks2_256 CRIS_GEMM_KS2=1 CRIS_GEMM_KS2_MAX_BLOCKS=256
echo "=== step"; cat $L.ab.log; tail -2 $L.ks2.err | cut -c1-300
