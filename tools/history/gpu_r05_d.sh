#!/bin/bash
# Round 5, call D: what the driver runs at round end (smoke, `pytest -m gpu`, the default bench line) plus the profiles/ evidence at
# this code state: kernel trace, memory-side traffic (two PMC passes), SQ counters, per-shape table, the other BASELINE configurations
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r05d
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | grep -v "$F" | tail -5 | cut -c1-300 > $L.smoke.log; cat $L.smoke.log
( time timeout 1190 python -m pytest tests/ -x -q -m gpu --durations=14 -p no:cacheprovider ) 2>&1 | grep -v "$F" | tail -34 | cut -c1-220 > $L.gpu_suite.log; tail -28 $L.gpu_suite.log
timeout 900 python bench.py --shape-table $L.gemm_shapes.tsv > $L.bench_n1.json 2>$L.bench.err; cut -c1-1200 $L.bench_n1.json; tail -3 $L.bench.err | cut -c1-200
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r05 -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-module-path > $L.prof.log 2>&1
echo "prof rc=$?"; grep '"metric"' $L.prof.log | cut -c1-200
db=$(find gpurun_out/prof -name "*_results.db" | head -1)
python tools/prof_summary.py $db $L.kernel_stats.csv 40 "void adam_kernel<1>" | tail -8
rm -rf gpurun_out/prof
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_$c; rm -rf $d
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $d -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-module-path --launch eager > $L.pmc_$c.log 2>&1
  echo "$c rc=$?"
done
python tools/pmc_summary.py $(find gpurun_out/pmc_FETCH_SIZE -name "*_results.db" | head -1) $(find gpurun_out/pmc_WRITE_SIZE -name "*_results.db" | head -1) $L.hbm_traffic.json 5 | tail -16
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
d=gpurun_out/pmc_sq; rm -rf $d
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $d -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-module-path --launch eager > $L.pmc_sq.log 2>&1
echo "sq rc=$?"; python tools/pmc_sq_summary.py $(find $d -name "*_results.db" | head -1) $L.sq_counters.json | tail -14; rm -rf $d
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timer --no-module-path"
x() { tag=$1; shift; timeout 300 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[0]); print('$tag', round(d['ms_per_step'],3), 'ms/step', round(d['value'],1), 'samples/s', d['step_roofline'])" >> $L.extra.log 2>&1; }
: > $L.extra.log
x r101_416 --spec r101
x r50_480_L22 --size 480
x r50_416_b16 --batch 16
x r50_416_b32 --batch 32
cat $L.extra.log
