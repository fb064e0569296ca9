#!/bin/bash
# (the code this exercises was REMOVED after calls L / M measured it slower: profiles/r04/call_l_m_staged_adam_rejected.patch holds it)
# Round 4, call L (the last 7 GPU-minutes): the per-stage optimizer update (CRIS_ADAM_STAGED) - bit-identity test, step time A/B,
# the multi-rank code paths with it on.  Every piece under its own timeout; most important first.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r04l
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
( time timeout 110 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "staged_optimizer" ) 2>&1 | grep -v "$F" | tail -12 | cut -c1-300 > $L.staged_test.log; cat $L.staged_test.log
B="--steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timer"
for rep in 1 2; do for s in 0 1; do
  CRIS_ADAM_STAGED=$s timeout 60 python bench.py $B 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('staged $s rep $rep: %.3f ms/step  %.1f samples/s  final_loss %.6f  %s' % (d['ms_per_step'], d['value'], d['config']['final_loss'], d['config']['optimizer_update']))" | tee -a $L.ab.log
done; done
( time CRIS_ADAM_STAGED=1 timeout 100 python -m pytest tests/test_dist_gpu.py -x -q -m gpu -k "rccl_collectives_inside or (two_ranks_equal and eager)" ) 2>&1 | grep -v "$F" | tail -8 | cut -c1-300 > $L.dist_tests.log; cat $L.dist_tests.log
for s in 0 1; do CRIS_ADAM_STAGED=$s timeout 45 python tools/dist1_check.py graph 40 2>&1 | grep "^mode" | sed "s/^/staged $s: /" | cut -c1-260 | tee -a $L.dist1.log; done
