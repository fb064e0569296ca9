#!/bin/bash
# round 3, call E: deep-ring variants, selection rules, module-path bench, multi-rank tests with the mailbox default, inference-only engine
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03e
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 400 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "8wave or deep_ring or conv_gemm_plain" 2>&1 | grep -v "$F" | tail -8 | cut -c1-400 > $L.kernel_tests.log
echo "=== kernel tests"; cat $L.kernel_tests.log
timeout 500 python tools/gemm_variants.py --min-m 1000 --min-k 256 --rounds 3 --variants 128x128,64x128,64x64,64x64d,64x128d,8w128x128 --tsv $L.variants.tsv 2>&1 | grep "GEMMVAR\|Error\|error" | cut -c1-300 > $L.variants.log
echo "=== variants"; cat $L.variants.log
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 200 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run off CRIS_GEMM8=0 CRIS_GEMM_NARROW_128=1
run dflt X=1
run deep1024 CRIS_GEMM_DEEP_MIN_K=1024
run deep2048 CRIS_GEMM_DEEP_MIN_K=2048
run dflt2 X=1
echo "=== step A/B"; cat $L.ab.log; tail -3 $L.dflt.err | cut -c1-300
timeout 300 python bench.py --path module --steps 50 --warmup 5 > $L.module.json 2> $L.module.err
echo "=== module path"; cut -c1-900 $L.module.json; tail -3 $L.module.err | cut -c1-300
timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_p2p_gpu.py tests/test_ref_loop_gpu.py tests/test_bench_launch.py tests/test_infer_gpu.py tests/test_module_gpu.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -8 | cut -c1-400 > $L.tests.log
echo "=== dist / infer / module tests"; cat $L.tests.log
