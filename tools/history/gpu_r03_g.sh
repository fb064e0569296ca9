#!/bin/bash
# round 3, call G: what the fixed cost of a tile is made of (probe: no stores / no statistics), selection fix check, module path
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03g
python -c "import __graft_entry__ as g; g.build()" || exit 1
: > $L.probe.log
for shape in "3 8 104 64 256 1" "3 8 26 512 512 1" "0 8 104 256 512 3" "2 8 104 64 256 1"; do
  for b in tools/probe/gemm8_probe_0 tools/probe/gemm8_probe_15 tools/probe/gemm8_probe_4; do
    for mode in 0 1 2; do
      echo -n "$(basename $b) " >> $L.probe.log
      timeout 60 $b $shape 20 $mode 2>&1 | grep G8PROBE >> $L.probe.log || echo >> $L.probe.log
    done
  done
done
echo "=== probes"; cat $L.probe.log
timeout 300 python bench.py --path module --steps 100 --warmup 5 > $L.module.json 2> $L.module.err
echo "=== module path"; cut -c1-330 $L.module.json; tail -2 $L.module.err | cut -c1-300
timeout 400 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --shape-table $L.shapes.tsv > $L.bench.json 2> $L.bench.err
echo "=== bench"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03g.bench.json').read())
print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['frac'], {k:(round(v['ms_per_step'],3),round(v['tflops'],1)) for k,v in d['kernels'].items()})
PY
timeout 300 python -m pytest tests/test_module_gpu.py -m gpu -q -x 2>&1 | tail -3
