#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03n
python -c "import __graft_entry__ as g; g.build()" || exit 1
CRIS_WGRAD8=1 timeout 300 python tools/wgrad_repro.py 2>&1 | grep REPRO | cut -c1-400 > $L.repro1.log
CRIS_WGRAD8=0 timeout 300 python tools/wgrad_repro.py 2>&1 | grep REPRO | cut -c1-400 > $L.repro0.log
echo "=== wgrad8 on"; cat $L.repro1.log; echo "=== off"; cat $L.repro0.log
