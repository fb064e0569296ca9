#!/bin/bash
# round 2, call E: small deterministic kernels reworked (bce, dmul, finalize), metric off the critical path; tile-variant A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02e
timeout 600 python -m pytest tests/test_hip_ops.py -q 2>&1 | tail -8 > $L.hip_ops.log
timeout 600 python -m pytest tests/test_engine_gpu.py -q -k "tiny_step or deterministic" 2>&1 | tail -6 > $L.engine.log
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 $B --shape-table $L.shapes_$tag.tsv 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), d['config']['final_loss'], {k: round(v['ms_per_step'],3) for k,v in d['kernels'].items()})" >> $L.ab.log 2>&1; }
: > $L.ab.log
run base X=1
run no128 CRIS_GEMM_T128_MIN=100000000
run all64x128 CRIS_GEMM_T128_MIN=100000000 CRIS_GEMM_T64_MAX=0
run ln512 CRIS_LN_BWD_BLOCKS=512
run bnred256 CRIS_BN_RED_BLOCKS=256
run wgblk768 CRIS_WGRAD_BLOCKS=768
for f in hip_ops engine ab; do echo "=== $f"; tail -12 $L.$f.log | cut -c1-600; done
