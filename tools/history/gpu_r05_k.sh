#!/bin/bash
# Round 5, call K (last): final library (8-wave 128x128 tile for the N 514 convolution): GEMM + engine tests, the default bench
# line and the kernel trace of this state
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r05k
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | grep -v "$F" | tail -4 | cut -c1-300
T() { tag=$1; shift; ( time timeout 900 python -m pytest "$@" -q -x -p no:cacheprovider ) 2>&1 | grep -v "$F" | tail -6 | cut -c1-200 > $L.$tag.log; echo "=== $tag"; tail -5 $L.$tag.log; }
T kernels tests/test_hip_ops.py -m gpu -k "gemm or conv"
T engine tests/test_engine_gpu.py -m gpu -k "tiny or small or config1 or deterministic or r101 or long_text or two_streams"
( time timeout 900 python bench.py --shape-table $L.gemm_shapes.tsv ) > $L.bench_n1.json 2>$L.bench.err; echo "bench stdout lines: $(wc -l < $L.bench_n1.json)"; cut -c1-400 $L.bench_n1.json
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r05 -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-module-path > $L.prof.log 2>&1
echo "prof rc=$?"
db=$(find gpurun_out/prof -name "*_results.db" | head -1)
python tools/prof_summary.py $db $L.kernel_stats.csv 40 "void adam_kernel<1>" | tail -4
rm -rf gpurun_out/prof
