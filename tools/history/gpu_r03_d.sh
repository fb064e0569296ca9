#!/bin/bash
# round 3, call D: tile order, 8w128x128 deep ring, automatic selection on: kernel tests, probes, per-shape table, step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03d
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 400 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "8wave or conv_gemm_plain or folded" 2>&1 | grep -v "$F" | tail -15 | cut -c1-400 > $L.kernel_tests.log
echo "=== kernel tests"; cat $L.kernel_tests.log
: > $L.probe.log
for shape in "0 8 104 256 512 3" "0 8 52 512 512 3" "2 8 104 512 256 3" "2 8 52 512 256 3" "3 8 26 512 512 3" "3 8 26 512 512 1" "3 8 13 2048 2048 1" "3 8 26 256 1024 1" "2 8 26 512 1024 3"; do
  for b in tools/probe/gemm8_probe_*; do
    echo -n "$(basename $b) " >> $L.probe.log
    timeout 60 $b $shape 20 2>&1 | grep G8PROBE >> $L.probe.log || echo >> $L.probe.log
  done
done
echo "=== probes"; cat $L.probe.log
timeout 500 python tools/gemm_variants.py --min-m 1000 --min-k 256 --rounds 3 --variants 128x128,64x128,64x64,8w256x256,8w128x256,8w128x128 --tsv $L.variants.tsv 2>&1 | grep "GEMMVAR\|Error\|error" | cut -c1-300 > $L.variants.log
echo "=== variants"; cat $L.variants.log
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 200 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run off CRIS_GEMM8=0
run on CRIS_GEMM8=1
run on_t128 CRIS_GEMM8=1 CRIS_GEMM8_T128_LO=100 CRIS_GEMM8_T128_HI=260
run on_t128b CRIS_GEMM8=1 CRIS_GEMM8_T128_LO=100 CRIS_GEMM8_T128_HI=360
run off2 CRIS_GEMM8=0
echo "=== step A/B"; cat $L.ab.log; tail -3 $L.on.err | cut -c1-300
