#!/bin/bash
# round 3, call I: fast lean epilogue: all GEMM kernel tests, probes (fast vs general epilogue builds), per-shape table, step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03i
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm or dgrad or pack" 2>&1 | grep -v "$F" | tail -8 | cut -c1-500 > $L.kernel_tests.log
echo "=== kernel tests"; cat $L.kernel_tests.log
: > $L.probe.log
for shape in "3 8 104 64 256 1" "3 8 26 512 512 1" "3 8 26 512 512 3" "0 8 104 256 512 3" "2 8 52 512 256 3"; do
  for b in tools/probe/gemm8_probe_*; do
    echo -n "$(basename $b) " >> $L.probe.log
    timeout 60 $b $shape 20 0 2>&1 | grep G8PROBE >> $L.probe.log || echo >> $L.probe.log
  done
done
echo "=== probes"; cat $L.probe.log
timeout 500 python tools/gemm_variants.py --min-m 1000 --rounds 3 --variants 128x128,64x128,64x64,128x64,8w256x256,8w128x256,8w128x128,stream128,stream64 --tsv $L.variants.tsv 2>&1 | grep "GEMMVAR" | cut -c1-300 > $L.variants.log
echo "=== variants"; cat $L.variants.log
timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --shape-table $L.shapes.tsv > $L.bench.json 2> $L.bench.err
echo "=== bench"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03i.bench.json').read())
print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['frac'], {k:(round(v['ms_per_step'],3),round(v['tflops'],1)) for k,v in d['kernels'].items()})
PY
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "not trajectory and not teacher" 2>&1 | grep -v "$F" | tail -5 | cut -c1-400
