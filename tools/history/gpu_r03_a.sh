#!/bin/bash
# round 3, call A: the 8-wave GEMM tiles (kernel tests, per-shape variant A/B, step A/B), parity by measurement (oracle on
# the GPU: fp32 / autocast, teacher-forced test), multi-rank launch + p2p tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03a
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 400 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "8wave or variant_refused or conv_gemm_plain" 2>&1 | grep -v "$F" | tail -15 | cut -c1-400 > $L.kernel_tests.log
echo "=== kernel tests"; cat $L.kernel_tests.log
timeout 500 python tools/gemm_variants.py --min-m 5000 --rounds 5 --tsv $L.variants.tsv 2>&1 | grep "GEMMVAR\|Error\|error" | cut -c1-400 > $L.variants.log
echo "=== variants"; cat $L.variants.log
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 200 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run off CRIS_GEMM8_MIN_TILES=0
run g8_150 CRIS_GEMM8_MIN_TILES=150
run g8_300 CRIS_GEMM8_MIN_TILES=300
run off2 CRIS_GEMM8_MIN_TILES=0
echo "=== step A/B"; cat $L.ab.log; tail -3 $L.g8_150.err | cut -c1-300
timeout 600 python tools/parity_study.py --steps 100 --out $L.parity.json 2>&1 | grep "PARITY\|Error\|error" | cut -c1-600 > $L.parity.log
echo "=== parity study"; cat $L.parity.log
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_oracle_device.py -m gpu -q -x -s -k "teacher_forced or oracle_on_the_gpu" 2>&1 | grep -v "$F" | tail -12 | cut -c1-600 > $L.teacher.log
echo "=== teacher forced"; cat $L.teacher.log
timeout 600 python -m pytest tests/test_bench_launch.py tests/test_p2p_gpu.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -8 | cut -c1-400 > $L.dist.log
echo "=== launch + p2p"; cat $L.dist.log
