#!/bin/bash
# round 2, call AD: drop-in module eval on the folded engine; smoke + default bench line at HEAD; kernel trace of the inference path
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02ad
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids"
timeout 300 python -m pytest tests/test_module_gpu.py tests/test_eval_post.py tests/test_engine_gpu.py -m gpu -q -k "module_eval or dataparallel or eval_runs_folded or eval_post or validate or tiny_step or deterministic" 2>&1 | grep -v "$F" | tail -25 | cut -c1-500 > $L.tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $L.smoke.log 2>&1; echo "smoke rc=$?" >> $L.smoke.log
timeout 400 python bench.py > $L.bench.json 2> $L.bench.err
rm -rf gpurun_out/prof
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o infer -- python tools/latency.py --iters 100 --batches 1,32 --modes fold+graph > $L.prof.log 2>&1
db=$(find gpurun_out/prof -name "*.db" | head -1); python tools/prof_summary.py $db $L.infer_kernel_stats.csv 360 > $L.prof_summary.log 2>&1
rm -rf gpurun_out/prof
echo "=== tests"; cat $L.tests.log
echo "=== smoke"; tail -2 $L.smoke.log | cut -c1-300
echo "=== bench"; cut -c1-900 $L.bench.json; tail -2 $L.bench.err | cut -c1-300
echo "=== prof"; head -14 $L.prof_summary.log | cut -c1-180; grep LATENCY $L.prof.log | cut -c1-600
