#!/bin/bash
# round 3, call AC (final state): module / distributed / inference / p2p tests, smoke, default bench, kernel trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03ac
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python -m pytest tests/test_module_gpu.py tests/test_infer_gpu.py tests/test_p2p_gpu.py tests/test_dist_gpu.py tests/test_ref_loop_gpu.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -4 | cut -c1-300 > $L.tests.log
echo "=== tests"; cat $L.tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $L.smoke.log 2>&1; echo "smoke rc=$?" >> $L.smoke.log; tail -2 $L.smoke.log | cut -c1-300
timeout 400 python bench.py --shape-table $L.shapes.tsv > $L.bench.json 2> $L.bench.err
echo "=== bench"; cut -c1-330 $L.bench.json
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r03ac -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timer > $L.prof.log 2>&1
python tools/prof_summary.py $(ls gpurun_out/prof/*/r03ac_results.db gpurun_out/prof/r03ac_results.db 2>/dev/null | head -1) $L.kernel_stats.csv 40 "void adam_kernel<1>" 2>&1 | tail -3
rm -rf gpurun_out/prof
grep "conv_gemm8_kernel<2, 2, 1>" $L.kernel_stats.csv | awk -F, '{print $NF, $(NF-1), $(NF-5), substr($1,1,50)}'
