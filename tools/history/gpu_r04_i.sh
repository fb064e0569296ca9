#!/bin/bash
# Round 4, call I: the tests that moved / changed after call H, and two tile-threshold arms
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r04i
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
T() { tag=$1; shift; timeout 900 "$@" 2>&1 | grep -v "$F" | tail -12 | cut -c1-400 > $L.$tag.log; echo "=== $tag"; tail -6 $L.$tag.log; }
T p2p python -m pytest tests/test_p2p_gpu.py -m gpu -q -x -p no:cacheprovider
T long_configs python -m pytest tests/test_parity_long_gpu.py -m gpu_long -q -x -p no:cacheprovider -k "config3 or config4"
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 300 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run base X=1
run t128_lo80 CRIS_GEMM8_T128_LO=80
run t128_lo60 CRIS_GEMM8_T128_LO=60
run t128_hi350 CRIS_GEMM8_T128_HI=350
run base2 X=1
cat $L.ab.log
