#!/bin/bash
# Round 5, call P: the one-rank DDP child command alone, three steps, stdout the plain way (was the abort of call r05o the re-pointed fd 1 or the short run?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 45 python bench.py --path module --ddp-one-rank --optimizer torch --steps 3 --warmup 5 > gpurun_out/r05p.out.txt 2> gpurun_out/r05p.err.txt; echo "rc=$? json lines: $(grep -c '^{' gpurun_out/r05p.out.txt)"; tail -3 gpurun_out/r05p.err.txt | cut -c1-200
