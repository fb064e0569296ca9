#!/bin/bash
# Round 5, call C: weight gradients on a stream of their own underneath the dgrad / BatchNorm chain (CRIS_WGRAD_STREAM=1): parity
# tests with the switch on, then the step A/B; and once more the two round-4 features that measured level (grouped GEMM launches,
# BatchNorm-backward partials from the dgrad epilogue) - the data for keeping or deleting them
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r05c
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
T() { tag=$1; shift; ( time timeout 900 "$@" -q -x -p no:cacheprovider --durations=4 ) 2>&1 | grep -v "$F" | tail -30 | cut -c1-400 > $L.$tag.log; echo "=== $tag"; tail -10 $L.$tag.log; }
T engine_ws env CRIS_WGRAD_STREAM=1 python -m pytest tests/test_engine_gpu.py -m gpu -k "tiny or config1 or deterministic or two_streams"
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer --no-module-path"
run() { tag=$1; shift; timeout 300 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[0]); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'), d['config'].get('graph_captured'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run base X=1
run wstream CRIS_WGRAD_STREAM=1
run nogroups CRIS_GEMM_GROUPS=0
run nobnr CRIS_BNR_FUSE=0
run base2 X=1
run wstream2 CRIS_WGRAD_STREAM=1
run nogroups2 CRIS_GEMM_GROUPS=0
run nobnr2 CRIS_BNR_FUSE=0
run neither CRIS_GEMM_GROUPS=0 CRIS_BNR_FUSE=0
echo "=== step A/B"; cat $L.ab.log; tail -3 $L.wstream.err | cut -c1-300
