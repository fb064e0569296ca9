#!/bin/bash
# Round 5, call J: what the per-shape variant table of call I suggests, in the step: the K-split 64x64 tile for few-block long-K
# problems (M 1352 layers, M 5408 / N 256), the 8-wave 128x128 tile for the N 514 convolution
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r05j
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
( time timeout 600 env CRIS_GEMM_KS2=1 CRIS_GEMM_KS2_MIN_K=1024 CRIS_GEMM_KS2_MAX_BLOCKS=352 CRIS_GEMM8_T128_HI=216 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "tiny or config1 or deterministic" ) 2>&1 | grep -v "$F" | tail -5 | cut -c1-200
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer --no-module-path"
run() { tag=$1; shift; timeout 300 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[0]); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run base X=1
run ks2_k1024 CRIS_GEMM_KS2=1 CRIS_GEMM_KS2_MIN_K=1024 CRIS_GEMM_KS2_MAX_BLOCKS=352
run ks2_k2048 CRIS_GEMM_KS2=1 CRIS_GEMM_KS2_MIN_K=2048 CRIS_GEMM_KS2_MAX_BLOCKS=352
run t128hi CRIS_GEMM8_T128_HI=216
run base2 X=1
run ks2_k1024b CRIS_GEMM_KS2=1 CRIS_GEMM_KS2_MIN_K=1024 CRIS_GEMM_KS2_MAX_BLOCKS=352
run ks2_k2048b CRIS_GEMM_KS2=1 CRIS_GEMM_KS2_MIN_K=2048 CRIS_GEMM_KS2_MAX_BLOCKS=352
run t128hib CRIS_GEMM8_T128_HI=216
run all CRIS_GEMM_KS2=1 CRIS_GEMM_KS2_MIN_K=1024 CRIS_GEMM_KS2_MAX_BLOCKS=352 CRIS_GEMM8_T128_HI=216
echo "=== step A/B"; cat $L.ab.log
