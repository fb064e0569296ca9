#!/bin/bash
# round 3, call J: split-K skinny kernels (tests, standalone timing, step A/B), engine tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03j
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 400 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "skinny or epilogues or transposed or plain" 2>&1 | grep -v "$F" | tail -8 | cut -c1-600 > $L.kernel_tests.log
echo "=== kernel tests"; cat $L.kernel_tests.log
timeout 200 python tools/skinny_bench.py 2>&1 | grep SKINNY > $L.skinny.log
echo "=== skinny"; cat $L.skinny.log
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 200 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run split CRIS_SKINNY_SPLIT=1
run old CRIS_SKINNY_SPLIT=0
run split128 CRIS_SKINNY_SPLIT=1 CRIS_SKINNY_BLOCKS=128
run split384 CRIS_SKINNY_SPLIT=1 CRIS_SKINNY_BLOCKS=384
run old2 CRIS_SKINNY_SPLIT=0
echo "=== step A/B"; cat $L.ab.log; tail -3 $L.split.err | cut -c1-300
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "tiny or r50_small or deterministic or stage" 2>&1 | grep -v "$F" | tail -5 | cut -c1-400
