#!/bin/bash
# (the code this exercises was REMOVED after this call measured it slower: profiles/r04/call_l_m_staged_adam_rejected.patch holds it)
# Round 4, call M (the last GPU seconds): the per-stage optimizer update with BOUNDED launches (cris_adam_step_bounded) - kernel
# bit-identity, trainer bit-identity, step time against the single pass for three grid limits.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r04m
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
( time timeout 70 python -m pytest tests/test_hip_ops.py tests/test_engine_gpu.py -x -q -m gpu -k "adam or staged_optimizer" ) 2>&1 | grep -v "$F" | tail -8 | cut -c1-300 > $L.tests.log; cat $L.tests.log
B="--steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timer"
run() {  # staged blocks label
  CRIS_ADAM_STAGED=$1 CRIS_ADAM_STAGED_BLOCKS=$2 timeout 50 python bench.py $B 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('staged $1 blocks $2: %.3f ms/step  %.1f samples/s  final_loss %.6f' % (d['ms_per_step'], d['value'], d['config']['final_loss']))" | tee -a $L.ab.log
}
run 0 0; run 1 256; run 1 128; run 0 0; run 1 256; run 1 512; run 1 64
