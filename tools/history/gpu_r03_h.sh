#!/bin/bash
# round 3, call H: streaming GEMM kernel (tests, per-shape table on the K <= 256 layers, step A/B), lean-only 8-wave selection
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03h
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 400 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "streaming or 8wave_tiles or conv_gemm_plain" 2>&1 | grep -v "$F" | tail -8 | cut -c1-500 > $L.kernel_tests.log
echo "=== kernel tests"; cat $L.kernel_tests.log
timeout 400 python - <<'PY' 2>&1 | grep "GEMMVAR" | cut -c1-260 > $L.variants.log
import sys; sys.argv=["x","--min-m","20000","--rounds","3","--variants","128x128,64x128,64x64,128x64,stream128,stream64"]
sys.path.insert(0,"tools")
import gemm_variants as G
# only the short-K shapes (K <= 256) incl. those below the tool's default K filter
orig=G.shapes_of_step
G.shapes_of_step=lambda: [s for s in orig() if s[2] <= 256 and s[3] == 1]
G.main()
PY
echo "=== variants (K <= 256)"; cat $L.variants.log
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 200 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run dflt X=1
run stream20k CRIS_GEMM_STREAM_MIN_M=20000
run stream20k_b512 CRIS_GEMM_STREAM_MIN_M=20000 CRIS_GEMM_STREAM_BLOCKS=512
run stream20k_b1024 CRIS_GEMM_STREAM_MIN_M=20000 CRIS_GEMM_STREAM_BLOCKS=1024
run dflt2 X=1
echo "=== step A/B"; cat $L.ab.log; tail -3 $L.stream20k.err | cut -c1-300
