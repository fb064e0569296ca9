#!/bin/bash
# Next round, first perf A/B (DESIGN.md "Where the GEMM-shaped time goes"): the 256x128 block / 128x64 wave-tile variant of
# conv_gemm_kernel with a 3-deep ring on the largest layers.  Applies tools/next_round/gemm_256x128.patch ON THE GPU BOX ONLY
# (the tree is unchanged), rebuilds there, checks the kernel against torch on the shapes that select it, then times the step
# with the variant off and on at three thresholds.   gpurun --timeout 900 -- 'bash tools/next_round/ab_256x128.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/ab256
git apply tools/next_round/gemm_256x128.patch 2>/dev/null || patch -p1 < tools/next_round/gemm_256x128.patch || { echo "patch does not apply"; exit 1; }
python -c "import __graft_entry__ as g; g.build()" || exit 1
F="Warning\|warn\|amdgpu.ids"
CRIS_GEMM_T256_MIN=1 timeout 300 python -m pytest tests/test_hip_ops.py -m gpu -q -k "conv_gemm" 2>&1 | grep -v "$F" | tail -6 | cut -c1-300 > $L.kernel_tests.log
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 120 env "$@" $B --shape-table $L.$tag.tsv 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run off CRIS_GEMM_T256_MIN=0
run min300 CRIS_GEMM_T256_MIN=300
run min160 CRIS_GEMM_T256_MIN=160
run min80 CRIS_GEMM_T256_MIN=80
run off2 CRIS_GEMM_T256_MIN=0
echo "=== kernel tests (variant forced wherever N >= 128)"; cat $L.kernel_tests.log
echo "=== step time"; cat $L.ab.log
echo "=== largest shapes, off vs min160"; for t in off min160; do echo $t; grep "M86528\|M21632" $L.$t.tsv | sort -t$'\t' -k4 -nr | head -8; done
