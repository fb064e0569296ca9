#!/bin/bash
# Round 5, call E: the small kernels around the GEMMs - dynamic-convolution backward reading x once per pixel, reciprocal index
# arithmetic in the pooling / resampling / stem kernels, forward dynamic convolution with more pixels per block: unit tests, engine
# parity, step A/B against the library of call D (variant prevd)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r05e
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
T() { tag=$1; shift; ( time timeout 900 python -m pytest "$@" -q -x -p no:cacheprovider --durations=3 ) 2>&1 | grep -v "$F" | tail -14 | cut -c1-300 > $L.$tag.log; echo "=== $tag"; tail -8 $L.$tag.log; }
T kernels tests/test_hip_ops.py -m gpu -k "pool or upsample or dynconv or stem or bn_apply"
T engine tests/test_engine_gpu.py -m gpu -k "tiny or config1 or other_shapes or deterministic"
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer --no-module-path"
run() { tag=$1; shift; timeout 300 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[0]); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run prevd CRIS_LIB_VARIANT=prevd
run new X=1
run new_ppb128 CRIS_DYNCONV_PPB=128
run new_ppb256 CRIS_DYNCONV_PPB=256
run prevd2 CRIS_LIB_VARIANT=prevd
run new2 X=1
run new_ppb128b CRIS_DYNCONV_PPB=128
run new_ppb256b CRIS_DYNCONV_PPB=256
echo "=== step A/B"; cat $L.ab.log
# the bench line's stdout must be ONE line (the RCCL banner of the one-rank DDP runs goes to stderr)
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-timer > $L.bench_stdout.txt 2>/dev/null; echo "bench stdout lines: $(wc -l < $L.bench_stdout.txt)"; cut -c1-200 $L.bench_stdout.txt | head -3
