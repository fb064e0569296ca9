#!/bin/bash
# round 2, call V: weight-gradient split / flush knobs (env only, one box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02v.log; : > $L
bench() { timeout 300 python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value'],1), d['config']['final_loss'])" >> $L 2>&1; }
for e in "X=1" "CRIS_WGRAD_MIN_STEPS=4" "CRIS_WGRAD_MIN_STEPS=6" "CRIS_WGRAD_MIN_STEPS=12" "CRIS_WGRAD_BLOCKS=384" "CRIS_WGRAD_BLOCKS=768" "CRIS_WGRAD_FLUSH_BLOCKS=2048" "CRIS_WGRAD_FLUSH_BLOCKS=6000" "X=2"; do
  echo "### env $e" >> $L; export $e; bench; unset ${e%%=*}
done
cat $L
