#!/bin/bash
# Round 4, call B: first GPU run of the grouped GEMM launches + fast general epilogue, the in-kernel SyncBN exchange, the module
# fixes (gradient accumulation, superseded forward) and the optional fused Adam
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r04b
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
T() { tag=$1; shift; timeout 900 python -m pytest "$@" -q -x -p no:cacheprovider 2>&1 | grep -v "$F" | tail -25 | cut -c1-600 > $L.$tag.log; echo "=== $tag"; tail -12 $L.$tag.log; }
T kernels tests/test_hip_ops.py -m gpu -k "conv_gemm or skinny or layernorm or wgrad or bn_ or syncbn or adam or dgrad"
T engine tests/test_engine_gpu.py -m gpu -k "tiny or config1 or deterministic or two_streams or stage_isolated or eval_forward or other_shapes"
T p2p tests/test_p2p_gpu.py tests/test_dist_gpu.py tests/test_comm_gpu.py -m gpu
T module tests/test_module_gpu.py tests/test_ref_loop_gpu.py tests/test_infer_gpu.py -m gpu
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 300 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'), d['config'].get('gpu_state_end'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run base X=1
run nogroups CRIS_GEMM_GROUPS=0
run f4auto CRIS_GROUP_VARIANT_F4_PROJ=auto
run f4_64x128 CRIS_GROUP_VARIANT_F4_PROJ=64x128
run base2 X=1
echo "=== step A/B"; cat $L.ab.log
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --shape-table $L.shapes.tsv > $L.bench.json 2>$L.bench.err; python -c "
import json; d=json.load(open('$L.bench.json')); print({k:d[k] for k in ('ms_per_step','value','roofline','step_roofline')}); print(d['kernels'])"
head -40 $L.shapes.tsv
for m in "CRIS_SYNCBN_P2P=0" "CRIS_SYNCBN_P2P=1 CRIS_SYNCBN_FUSED=0" "CRIS_SYNCBN_P2P=1 CRIS_SYNCBN_FUSED=1"; do
  timeout 300 env $m python tools/dist1_check.py graph 40 2>&1 | grep "^mode" | cut -c1-400 >> $L.dist1.log
done
echo "=== forced multi-rank code paths, one rank (plain step: see base above)"; cat $L.dist1.log
B2="python bench.py --path module --steps 60 --warmup 10 --no-cpu-baseline"
for o in torch cris; do timeout 300 $B2 --optimizer $o 2>$L.module_$o.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('module/$o', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('optimizer'), d['config'].get('final_loss'))" >> $L.module.log 2>&1; done
echo "=== module path"; cat $L.module.log
