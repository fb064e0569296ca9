#!/bin/bash
# round 2, call T: LDS-DMA ring depth per GEMM tile variant: rebuild on the box, bench each arm
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02t.log; : > $L
for arm in "" "-DST_64x64=4" "-DST_64x64=5" "-DST_128x128=3" "-DST_64x128=4" "-DST_128x64=4" ""; do
  export CRIS_EXTRA_HIPCC_FLAGS="$arm"
  ( time python -c "from cris.pytorch_amd.csrc import build; build.build()" ) 2>&1 | grep "real\|rror" >> $L
  echo "### arm '$arm'" >> $L
  timeout 300 python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value'],1), d['config']['final_loss'])" >> $L 2>&1
done
cat $L
