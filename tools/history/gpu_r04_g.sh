#!/bin/bash
# Round 4, call G: module path after moving the gradient hand-over to the start of backward (no per-step copies under the reference's
# loop order), graphs vs command lists; engine determinism / parity with the arena cleared in backward
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r04g
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
T() { tag=$1; shift; timeout 900 "$@" 2>&1 | grep -v "$F" | tail -25 | cut -c1-700 > $L.$tag.log; echo "=== $tag"; tail -8 $L.$tag.log; }
T module python -m pytest tests/test_module_gpu.py tests/test_ref_loop_gpu.py tests/test_infer_gpu.py -m gpu -q -x -p no:cacheprovider
T engine python -m pytest tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider -k "tiny or config1 or deterministic or two_streams or stage_isolated or sentence_vector or long_text"
T dist python -m pytest tests/test_dist_gpu.py -m gpu -q -x -p no:cacheprovider -k "not rccl"
B2="python bench.py --path module --steps 60 --warmup 10 --no-cpu-baseline --phase-times"
mp() { tag=$1; shift; timeout 300 env "$@" 2>$L.module_$tag.err | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('module/$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('optimizer'), d['config'].get('replay'))
for k,v in (d.get('phase_times') or {}).items(): print('   %-14s host %.2f ms  device %.2f ms' % (k, v['host_ms'], v['device_ms']))" >> $L.module_bench.log 2>&1; }
: > $L.module_bench.log
mp torch_graph X=1 $B2
mp torch_cmdlist CRIS_MODULE_REPLAY=cmdlist $B2
mp cris_graph X=1 $B2 --optimizer cris
mp cris_cmdlist CRIS_MODULE_REPLAY=cmdlist $B2 --optimizer cris
echo "=== module path"; cat $L.module_bench.log
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('native', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('gpu_state_end'))"
