#!/bin/bash
# round 2, call S: cache policy of the GEMM output stores (aux bits 0 / 2 / 1 / 3): rebuild on the box, bench each
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02s.log; : > $L
for aux in 2 1 17 0; do
  export CRIS_EXTRA_HIPCC_FLAGS="-DCRIS_ST_AUX=$aux"
  python -c "from cris.pytorch_amd.csrc import build; build.build()" >> $L 2>&1
  echo "### aux=$aux" >> $L
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['final_loss'])" >> $L 2>&1
done
cat $L
