#!/bin/bash
# SQ counters per kernel (where do the waves spend their cycles?) -> gpurun_out/pmc_sq/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
d=gpurun_out/pmc_sq; rm -rf $d
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $d -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --launch eager > gpurun_out/pmc_sq.log 2>&1
echo "rc=$?"; python tools/pmc_sq_summary.py $d/pmc_results.db gpurun_out/pmc_sq_summary.json; rm -rf $d
