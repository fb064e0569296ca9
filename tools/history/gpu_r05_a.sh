#!/bin/bash
# Round 5, call A: the new driver-suite members (float64-teacher parity runs, world 2/4/8 mailbox tests, fused-Adam advisor tests,
# 8-rank launch test) with their durations; the bench line with module_path + swept CPU baseline; phase stamps of the 64x64 tile;
# store-policy A/B (write-through / non-temporal epilogue stores); small-N/K weight-gradient sensitivity to the split count
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r05a
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
T() { tag=$1; shift; ( time timeout 1500 python -m pytest "$@" -q -x -p no:cacheprovider --durations=12 ) 2>&1 | grep -v "$F" | tail -40 | cut -c1-400 > $L.$tag.log; echo "=== $tag"; tail -22 $L.$tag.log; }
T parity tests/test_parity_long_gpu.py -m gpu -s
T module tests/test_module_gpu.py -m gpu -k "fused or optional"
T p2p tests/test_p2p_gpu.py -m gpu
T launch8 tests/test_bench_launch.py -m gpu -k eight
# probes (seconds each)
P=tools/probe
: > $L.probe.log
for sh in "8 26 512 512 1" "8 26 1024 256 1" "8 26 256 1024 1" "8 26 256 256 3" "8 13 512 512 3"; do
  for b in full nomfma nodma noepi st5 sc1; do timeout 60 $P/gemm4_probe_$b 64x64 $sh >> $L.probe.log 2>&1; done
  timeout 60 $P/gemm4_probe_full 64x128 $sh >> $L.probe.log 2>&1
  timeout 60 $P/gemm4_probe_full 128x128 $sh >> $L.probe.log 2>&1
done
echo "=== probe"; grep "^G4" $L.probe.log | head -60
# step A/B: epilogue / apply-kernel store policy
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer --no-module-path"
run() { tag=$1; shift; timeout 300 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run base X=1
run sc1 CRIS_LIB_VARIANT=sc1
run nt CRIS_LIB_VARIANT=nt
run base2 X=1
run sc1b CRIS_LIB_VARIANT=sc1
echo "=== step A/B (store policy)"; cat $L.ab.log
# weight gradients of the small-N/K shapes: sensitivity to the number of blocks (splits)
: > $L.wgrad.log
for nb in 512 768 1024 1536; do CRIS_WGRAD_BLOCKS=$nb timeout 200 python tools/wgrad_bench.py --small 2>&1 | grep WGRAD >> $L.wgrad.log; done
echo "=== wgrad small"; cat $L.wgrad.log | cut -c1-150
# the bench line as the driver runs it (module_path + CPU baseline inside)
( time timeout 900 python bench.py --steps 200 --warmup 10 ) 2>$L.bench.err | tee $L.bench_n1.json | cut -c1-1500; tail -4 $L.bench.err | cut -c1-200
