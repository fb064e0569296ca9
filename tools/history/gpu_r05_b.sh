#!/bin/bash
# Round 5, call B: kernels after the prologue change (reciprocal divisions, 1x1 fast path) - unit tests first; the repaired tests of
# call A; probes (new prologue against call A's 3020-tick median, the launch floor of the grid, prologue + epilogue only, A operand
# through registers with BatchNorm + ReLU against the DMA path); step A/B old prologue / new / deep-ring weight gradients; a second
# float64-teacher run of configs[1] on another box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r05b
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
T() { tag=$1; shift; ( time timeout 1500 python -m pytest "$@" -q -x -p no:cacheprovider --durations=6 ) 2>&1 | grep -v "$F" | tail -40 | cut -c1-400 > $L.$tag.log; echo "=== $tag"; tail -14 $L.$tag.log; }
T kernels tests/test_hip_ops.py -m gpu -k "gemm or conv or stem or wgrad or dynconv"
T engine tests/test_engine_gpu.py -m gpu -k "tiny or small or config1 or deterministic or other_shapes"
T module tests/test_module_gpu.py -m gpu -k "fused or optional" -s
T p2p tests/test_p2p_gpu.py -m gpu -k "trainer and 4"
P=tools/probe
: > $L.probe.log
for sh in "8 26 512 512 1" "8 26 256 1024 1" "8 26 256 256 3" "8 52 128 512 1" "8 52 128 128 3" "8 104 64 256 1" "8 104 64 64 3"; do
  for b in full empty noloop afuseref afuse; do timeout 60 $P/gemm4_probe_$b 64x64 $sh >> $L.probe.log 2>&1; done
done
echo "=== probe"; grep "^G4\|checksum\|phase 1\|lifetime" $L.probe.log | cut -c1-160 | head -120
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer --no-module-path"
run() { tag=$1; shift; timeout 300 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[0]); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run oldpro CRIS_LIB_VARIANT=oldpro
run new X=1
run deep CRIS_WGRAD_DEEP=1
run oldpro2 CRIS_LIB_VARIANT=oldpro
run new2 X=1
run deep2 CRIS_WGRAD_DEEP=1
echo "=== step A/B"; cat $L.ab.log
: > $L.wgrad.log
for d in 0 1; do CRIS_WGRAD_DEEP=$d timeout 200 python tools/wgrad_bench.py --small 2>&1 | grep WGRAD | sed "s/^/DEEP=$d /" >> $L.wgrad.log; done
echo "=== wgrad small (deep ring)"; cut -c1-130 $L.wgrad.log
T teacher tests/test_parity_long_gpu.py -m gpu -k "teacher_forced_r50_full" -s
cp gpurun_out/teacher_forced_r50.json gpurun_out/r05b.teacher_forced_r50_fp64.json 2>/dev/null
