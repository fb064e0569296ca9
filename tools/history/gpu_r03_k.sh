#!/bin/bash
# round 3, call K: is 15.0 ms of call J the box or the build?  step A/B again + kernel timer
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03k
python -c "import __graft_entry__ as g; g.build()" || exit 1
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 200 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run dflt X=1
run oldskinny CRIS_SKINNY_SPLIT=0
run g8off CRIS_GEMM8=0
run dflt2 X=1
echo "=== step A/B"; cat $L.ab.log
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --shape-table $L.shapes.tsv > $L.bench.json 2> $L.bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03k.bench.json').read())
print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['frac'], {k:(round(v['ms_per_step'],3),round(v['tflops'],1),v['launches_per_step']) for k,v in d['kernels'].items()})
PY
