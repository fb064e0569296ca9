#!/bin/bash
# round 3, call W: 32-row tiles for the 9-tap Adam / pack kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03w
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "adam or pack or skinny or 8wave" 2>&1 | grep -v "$F" | tail -5 | cut -c1-400 > $L.kernel_tests.log
echo "=== kernel tests"; cat $L.kernel_tests.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "not 100_steps" 2>&1 | grep -v "$F" | tail -4 | cut -c1-400 > $L.engine_tests.log
echo "=== engine tests"; cat $L.engine_tests.log
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 200 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run base X=1
run base2 X=1
echo "=== step"; cat $L.ab.log
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r03w -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timer > $L.prof.log 2>&1
python tools/prof_summary.py $(ls gpurun_out/prof/*/r03w_results.db gpurun_out/prof/r03w_results.db 2>/dev/null | head -1) $L.kernel_stats.csv 40 "void adam_kernel<1>" 2>&1 | tail -3
rm -rf gpurun_out/prof
grep "adam" $L.kernel_stats.csv | awk -F, '{print $NF, $(NF-1), $(NF-5), substr($1,1,50)}' | head
