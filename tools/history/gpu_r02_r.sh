#!/bin/bash
# round 2, call R: PMC passes at HEAD (HBM traffic, SQ counters), the multi-rank GPU tests incl. RCCL-in-graph
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02r
timeout 900 python -m pytest tests/test_dist_gpu.py -q -x 2>&1 | grep "passed\|failed\|Error" | tail -5 > $L.dist.log
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_$c; rm -rf $d
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $d -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --launch eager > $L.pmc_$c.log 2>&1
done
python tools/pmc_summary.py $(find gpurun_out/pmc_FETCH_SIZE -name "*.db" | head -1) $(find gpurun_out/pmc_WRITE_SIZE -name "*.db" | head -1) $L.hbm_traffic.json > $L.pmc_summary.log 2>&1
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
d=gpurun_out/pmc_sq; rm -rf $d
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $d -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --launch eager > $L.pmc_sq.log 2>&1
python tools/pmc_sq_summary.py $(find $d -name "*.db" | head -1) $L.sq_counters.json > $L.sq_summary.log 2>&1; rm -rf $d
echo "=== dist"; cat $L.dist.log
for f in pmc_summary sq_summary; do echo "=== $f"; tail -16 $L.$f.log | cut -c1-200; done
