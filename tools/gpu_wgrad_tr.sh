#!/bin/bash
# First GPU call for the experimental transposing-read weight-gradient kernel (gemm.hip: conv_wgrad_tr_kernel,
# CRIS_WGRAD_TR=1..3, default off): per-kernel parity for every mode, then the step time of each mode against the default.
#   gpurun --timeout 400 -- 'bash tools/gpu_wgrad_tr.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; out=gpurun_out/wgrad_tr.log; : > $out
for mode in 1 2 3; do
  echo "== parity CRIS_WGRAD_TR=$mode" >> $out
  CRIS_WGRAD_TR=$mode timeout 120 python -m pytest tests/test_hip_ops.py -q -x -k "wgrad or adam_gemm_layout" 2>&1 | tail -4 >> $out
done
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-kernel-timer"
for mode in 0 1 2 3; do
  echo "== bench CRIS_WGRAD_TR=$mode" >> $out
  CRIS_WGRAD_TR=$mode timeout 90 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['final_loss'])" >> $out 2>&1
done
cat $out
