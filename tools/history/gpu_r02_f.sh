#!/bin/bash
# round 2, call F: chunked BatchNorm backward (apply adds the partial rows itself), tile heuristics; A/B in one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02f
timeout 600 python -m pytest tests/test_hip_ops.py -q 2>&1 | tail -8 > $L.hip_ops.log
timeout 600 python -m pytest tests/test_engine_gpu.py -q -k "tiny_step or ragged or deterministic or stage_isolated or r50_small" 2>&1 | tail -8 > $L.engine.log
timeout 600 python -m pytest tests/test_dist_gpu.py tests/test_ref_loop_gpu.py -q 2>&1 | tail -6 > $L.dist.log
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 $B --shape-table $L.shapes_$tag.tsv 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), d['config']['final_loss'], {k: round(v['ms_per_step'],3) for k,v in d['kernels'].items()})" >> $L.ab.log 2>&1; }
: > $L.ab.log
run base X=1
run bnsum0 CRIS_BN_SUM_IN_APPLY=0
run base2 X=1
run bnsum0b CRIS_BN_SUM_IN_APPLY=0
for f in hip_ops engine dist ab; do echo "=== $f"; tail -10 $L.$f.log | cut -c1-500; done
