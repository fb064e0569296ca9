#!/bin/bash
# round 2, call U: knob sweep on one box (compile-time -D arms are rebuilt on the box; env arms reuse the default build)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02u.log; : > $L
bench() { timeout 300 python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value'],1), d['config']['final_loss'])" >> $L 2>&1; }
for arm in "" "-DST_64x128=2" "-DST_128x64=2" "-DWG_RUN=4" "-DWG_RUN=16" "-DADAM_ELEMS=4096" "-DADAM_ELEMS=16384" "-DSK_WAVES=4" ""; do
  export CRIS_EXTRA_HIPCC_FLAGS="$arm"
  python -c "from cris.pytorch_amd.csrc import build; build.build()" 2>&1 | grep -i "error" >> $L
  echo "### build '$arm'" >> $L; bench
done
export CRIS_EXTRA_HIPCC_FLAGS=""
python -c "from cris.pytorch_amd.csrc import build; build.build()" 2>&1 | grep -i "error" >> $L
for e in "X=1" "CRIS_GEMM_T128_MIN=320" "CRIS_GEMM_T128_MIN=700" "CRIS_GEMM_T128_MIN=1400" "CRIS_GEMM_T64_MAX=512" "CRIS_GEMM_T64_MAX=2048" "CRIS_BN_RED_BLOCKS=256" "CRIS_BN_RED_BLOCKS=1024" "CRIS_BN_RED_ROWS=16" "CRIS_BN_RED_ROWS=64" "CRIS_LN_BWD_BLOCKS=256" "CRIS_LN_BWD_BLOCKS=1024" "CRIS_ATTN_LDS_MIN=128" "CRIS_WGRAD_GROUP_M=4096" "CRIS_WGRAD_GROUP_M=22000" "X=2"; do
  echo "### env $e" >> $L; export $e; bench; unset ${e%%=*}
done
cat $L
