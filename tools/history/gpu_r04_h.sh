#!/bin/bash
# Round 4, call H: what the driver runs at round end - smoke, the whole `-m gpu` suite (with durations), then the long parity runs and a
# two-rank run of the bench at full size (two ranks sharing the one GPU, gloo carrying the host-side collectives)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r04h
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | grep -v "$F" | tail -6 | cut -c1-400 > $L.smoke.log; cat $L.smoke.log
( time timeout 1150 python -m pytest tests/ -x -q -m gpu --durations=15 ) 2>&1 | grep -v "$F" | tail -40 | cut -c1-300 > $L.gpu_suite.log; tail -30 $L.gpu_suite.log
( time timeout 900 python -m pytest tests/ -q -m gpu_long -s ) 2>&1 | grep -v "$F" | grep "teacher-forced\|passed\|failed\|real\|max |hip" | cut -c1-400 > $L.long.log; cat $L.long.log
timeout 600 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>$L.ddp2.err | cut -c1-1500 > $L.ddp2.json; cat $L.ddp2.json; tail -3 $L.ddp2.err | cut -c1-300
