#!/bin/bash
# round 3, call T: launch-geometry knob sweep on one box (each arm: 200 timed steps of the default bench)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03t
python -c "import __graft_entry__ as g; g.build()" || exit 1
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 200 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run base X=1
run bnred1024 CRIS_BN_RED_BLOCKS=1024
run bnred256 CRIS_BN_RED_BLOCKS=256
run lnbwd1024 CRIS_LN_BWD_BLOCKS=1024
run lnbwd256 CRIS_LN_BWD_BLOCKS=256
run wgblocks768 CRIS_WGRAD_BLOCKS=768
run wgblocks384 CRIS_WGRAD_BLOCKS=384
run wggroupm4096 CRIS_WGRAD_GROUP_M=4096
run wgflush2048 CRIS_WGRAD_FLUSH_BLOCKS=2048
run wgflush6144 CRIS_WGRAD_FLUSH_BLOCKS=6144
run skinny256 CRIS_SKINNY_BLOCKS=256
run skinny128 CRIS_SKINNY_BLOCKS=128
run g8tiles128 CRIS_GEMM8_MIN_TILES=128
run g8tiles200 CRIS_GEMM8_MIN_TILES=200
run t128hi256 CRIS_GEMM8_T128_HI=256
run w8blocks512 CRIS_WGRAD8_BLOCKS=512
run base2 X=1
cat $L.ab.log
