#!/bin/bash
# Round 4, call E: the profiles/ evidence at the final code state - kernel trace, memory-side traffic (two PMC passes), SQ counters,
# the default bench line with the CPU baseline, the other BASELINE configurations
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r04e
python -c "import __graft_entry__ as g; g.build()" || exit 1
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -v "^=\|^$" | head -12 > $L.smi_start.log
# 1. kernel trace: the last 40 steps of a traced run
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r04 -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timer > $L.prof.log 2>&1
echo "prof rc=$?"; grep '"metric"' $L.prof.log | cut -c1-200
db=$(find gpurun_out/prof -name "*_results.db" | head -1)
python tools/prof_summary.py $db $L.kernel_stats.csv 40 "void adam_kernel<1>" | tail -8
rm -rf gpurun_out/prof
# 2. memory-side traffic: FETCH_SIZE and WRITE_SIZE in separate passes (2 set-up + 1 warm-up + 2 timed steps = 5 steps, eager launches)
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_$c; rm -rf $d
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $d -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --launch eager > $L.pmc_$c.log 2>&1
  echo "$c rc=$?"
done
python tools/pmc_summary.py $(find gpurun_out/pmc_FETCH_SIZE -name "*_results.db" | head -1) $(find gpurun_out/pmc_WRITE_SIZE -name "*_results.db" | head -1) $L.hbm_traffic.json 5 | tail -16
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
# 3. SQ counters
d=gpurun_out/pmc_sq; rm -rf $d
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $d -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --launch eager > $L.pmc_sq.log 2>&1
echo "sq rc=$?"; python tools/pmc_sq_summary.py $(find $d -name "*_results.db" | head -1) $L.sq_counters.json | tail -14; rm -rf $d
# 4. the default bench line (500 timed steps, CPU baseline, per-shape table)
timeout 900 python bench.py --shape-table $L.gemm_shapes.tsv > $L.bench_n1.json 2>$L.bench.err; cut -c1-1500 $L.bench_n1.json
# 5. the other configurations
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timer"
x() { tag=$1; shift; timeout 300 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), 'ms/step', round(d['value'],1), 'samples/s', d['step_roofline'])" >> $L.extra.log 2>&1; }
: > $L.extra.log
x r101_416 --spec r101
x r50_480_L22 --size 480
x r50_416_b16 --batch 16
x r50_416_b32 --batch 32
cat $L.extra.log
# 6. the drop-in module under the reference's loop body (torch Adam = unchanged loop; cris = the optional one-line optimizer)
B2="python bench.py --path module --steps 100 --warmup 10 --no-cpu-baseline --phase-times"
mp() { tag=$1; shift; timeout 300 env "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('module/$tag', round(d['ms_per_step'],3), 'ms/step', round(d['value'],1), 'samples/s', d['config'].get('optimizer'), d['config'].get('replay'))
for k,v in (d.get('phase_times') or {}).items(): print('   %-14s host %.2f ms  device %.2f ms' % (k, v['host_ms'], v['device_ms']))" >> $L.module.log 2>&1; }
: > $L.module.log
mp torch_graph X=1 $B2
mp cris_graph X=1 $B2 --optimizer cris
mp cris_cmdlist CRIS_MODULE_REPLAY=cmdlist $B2 --optimizer cris
cat $L.module.log
timeout 300 python -m pytest tests/test_module_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -v "^=\|^$" | head -12 > $L.smi_end.log
