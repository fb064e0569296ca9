#!/bin/bash
# Round 4, call F: the fp32 sentence-vector path with the K-parallel kernel (step A/B), the B = 2 long-text case with it on / off,
# host profile of the module path's loop body
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r04f
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
T() { tag=$1; shift; timeout 900 "$@" 2>&1 | grep -v "$F" | tail -25 | cut -c1-700 > $L.$tag.log; echo "=== $tag"; tail -8 $L.$tag.log; }
T kernels python -m pytest tests/test_hip_ops.py -m gpu -q -x -p no:cacheprovider -k "small_fp32 or bn_backward_partials"
T longtext_f32 env CRIS_STATE_FP32=1 python -m pytest tests/test_engine_gpu.py -m gpu -q -s -p no:cacheprovider -k "long_text or r50_small"
T longtext_bf16 env CRIS_STATE_FP32=0 python -m pytest tests/test_engine_gpu.py -m gpu -q -s -p no:cacheprovider -k "long_text or r50_small"
T engine python -m pytest tests/test_engine_gpu.py tests/test_infer_gpu.py tests/test_module_gpu.py -m gpu -q -x -p no:cacheprovider -k "not long_text and not config3 and not config4 and not r101"
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 300 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run base X=1
run state_bf16 CRIS_STATE_FP32=0
run base2 X=1
run state_bf16_2 CRIS_STATE_FP32=0
echo "=== step A/B (sentence vector in fp32, K-parallel kernel)"; cat $L.ab.log
timeout 300 python bench.py --path module --optimizer cris --steps 30 --warmup 10 --no-cpu-baseline --pyprof 2> $L.pyprof.txt | cut -c1-300
grep -A 45 "Ordered by" $L.pyprof.txt | cut -c1-160
