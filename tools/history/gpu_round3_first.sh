#!/bin/bash
# FIRST gpurun call of the next round: the whole GPU suite at HEAD with its wall time (the round-2 session never ran it in one
# piece after the inference / JPEG / comm tests were added), smoke, the default bench line, and the measurement tools of the
# paths added in round 2.  About 6-8 GPU-minutes.   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_round3_first.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03a
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
t0=$(date +%s)
timeout 1300 python -m pytest tests -m gpu -q --durations=15 2>&1 | grep -v "$F" | tail -40 | cut -c1-300 > $L.gputests.log
echo "suite seconds: $(( $(date +%s) - t0 ))" >> $L.gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $L.smoke.log 2>&1; echo "smoke rc=$?" >> $L.smoke.log
timeout 400 python bench.py --shape-table $L.shapes.tsv > $L.bench.json 2> $L.bench.err
timeout 200 python tools/latency.py --iters 300 --modes fold+graph,nofold+graph 2>&1 | grep "LATENCY" | cut -c1-3000 > $L.latency.log
timeout 200 python tools/jpeg_bench.py --iters 30 --batch 64 --threads 32 2>&1 | grep "JPEGBENCH" | cut -c1-1500 > $L.jpeg.log
timeout 200 python tools/comm1_check.py r50 6 2>&1 | grep "COMM1" | cut -c1-1500 > $L.comm1.log
echo "=== gputests"; cat $L.gputests.log
echo "=== smoke"; tail -2 $L.smoke.log | cut -c1-300
echo "=== bench"; cut -c1-500 $L.bench.json; tail -2 $L.bench.err | cut -c1-200
echo "=== latency"; cat $L.latency.log
echo "=== jpeg"; cat $L.jpeg.log
echo "=== comm1"; cat $L.comm1.log
