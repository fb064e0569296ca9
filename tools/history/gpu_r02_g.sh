#!/bin/bash
# round 2, call G: the whole GPU test suite, then the committed measurements (kernel trace, HBM counters, SQ counters, bench)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02g
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "Warning\|warn\|return float\|Consider using\|amdgpu.ids" > $L.gputests.log
timeout 400 python bench.py --steps 200 --warmup 20 --shape-table $L.shapes.tsv > $L.bench.json 2> $L.bench.err
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timer > $L.prof.log 2>&1
db=$(find gpurun_out/prof -name "*.db" | head -1); python tools/prof_summary.py $db $L.kernel_stats.csv 44 > $L.prof_summary.log 2>&1
rm -rf gpurun_out/prof
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_$c; rm -rf $d
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $d -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --launch eager > $L.pmc_$c.log 2>&1
done
python tools/pmc_summary.py $(find gpurun_out/pmc_FETCH_SIZE -name "*.db" | head -1) $(find gpurun_out/pmc_WRITE_SIZE -name "*.db" | head -1) $L.hbm_traffic.json > $L.pmc_summary.log 2>&1
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
d=gpurun_out/pmc_sq; rm -rf $d
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $d -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --launch eager > $L.pmc_sq.log 2>&1
python tools/pmc_sq_summary.py $(find $d -name "*.db" | head -1) $L.sq_counters.json > $L.sq_summary.log 2>&1; rm -rf $d
echo "=== gputests"; tail -40 $L.gputests.log | cut -c1-400
for f in prof_summary pmc_summary sq_summary; do echo "=== $f"; tail -22 $L.$f.log | cut -c1-200; done
echo "=== bench"; cut -c1-2500 $L.bench.json; tail -3 $L.bench.err
