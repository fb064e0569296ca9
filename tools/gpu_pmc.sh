#!/bin/bash
# HBM traffic counters (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass) -> gpurun_out/pmc_{fetch,write}/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_$c; rm -rf $d
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $d -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --launch eager > gpurun_out/pmc_$c.log 2>&1
  echo "$c rc=$?"; ls -la $d | tail -3
done
