#!/bin/bash
# Round 4, call C: peer-mailbox tests (two ranks on one GPU), BatchNorm-backward partials from the dgrad epilogue, text-encoder
# error budget in detail, graph-launch latency probe
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r04c
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
T() { tag=$1; shift; timeout 900 python -m pytest "$@" -q -x -p no:cacheprovider 2>&1 | grep -v "$F" | tail -25 | cut -c1-600 > $L.$tag.log; echo "=== $tag"; tail -12 $L.$tag.log; }
T p2p tests/test_p2p_gpu.py -m gpu
T dist tests/test_dist_gpu.py tests/test_comm_gpu.py -m gpu
T kernels tests/test_hip_ops.py -m gpu -k "conv_gemm or bn_"
T engine tests/test_engine_gpu.py -m gpu -k "tiny or config1 or config3 or deterministic or two_streams or stage_isolated or first_12"
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 300 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run base X=1
run nobnr CRIS_BNR_FUSE=0
run base2 X=1
run nobnr2 CRIS_BNR_FUSE=0
echo "=== step A/B (BatchNorm-backward partials in the dgrad epilogue)"; cat $L.ab.log
timeout 300 python tools/graph_latency.py 100 2>&1 | grep LATENCY > $L.latency.log; cat $L.latency.log
timeout 600 python tools/error_budget.py --spec r50 --steps 100 --every 8 --detail text --no-hip --out gpurun_out/error_budget_r50_text.json 2>&1 | grep "BUDGET summary" -A 20 | cut -c1-200 > $L.budget_text.log; cat $L.budget_text.log
timeout 600 python tools/error_budget.py --spec r50 --steps 100 --every 8 --detail neck --no-hip --out gpurun_out/error_budget_r50_neck.json 2>&1 | grep "BUDGET summary" -A 20 | cut -c1-200 > $L.budget_neck.log; cat $L.budget_neck.log
