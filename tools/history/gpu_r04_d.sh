#!/bin/bash
# Round 4, call D: the sentence-vector path in fp32 (csrc/smallf32.hip) - kernel tests, engine / module / multi-rank tests, the
# 100-state teacher-forced run with it on (twice: is the deterministic teacher reproducible?) and off; module-path phase times
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r04d
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
T() { tag=$1; shift; timeout 900 python -m pytest "$@" -q -x -p no:cacheprovider 2>&1 | grep -v "$F" | tail -25 | cut -c1-600 > $L.$tag.log; echo "=== $tag"; tail -8 $L.$tag.log; }
T kernels tests/test_hip_ops.py -m gpu -k "small_fp32 or bn_backward_partials or syncbn"
T engine tests/test_engine_gpu.py -m gpu -k "tiny or r50_small or r101_step or long_text or config1 or deterministic or stage_isolated or eval_forward or other_shapes"
T module tests/test_module_gpu.py tests/test_infer_gpu.py tests/test_dist_gpu.py -m gpu
TF() { tag=$1; shift; timeout 600 env "$@" python -m pytest tests/test_parity_long_gpu.py -m gpu_long -q -s -k "teacher_forced_r50_full_size_100" -p no:cacheprovider 2>&1 | grep "teacher-forced\|passed\|failed" | cut -c1-400 > $L.tf_$tag.log; cp gpurun_out/teacher_forced_r50.json gpurun_out/teacher_forced_r50_$tag.json; echo "=== teacher-forced $tag"; cat $L.tf_$tag.log; }
TF f32_run1 CRIS_STATE_FP32=1
TF f32_run2 CRIS_STATE_FP32=1
TF bf16 CRIS_STATE_FP32=0
timeout 600 python -m pytest tests/test_parity_long_gpu.py -m gpu_long -q -s -k "r101_20 or 480_22" -p no:cacheprovider 2>&1 | grep "teacher-forced\|passed\|failed" | cut -c1-400 > $L.tf_other.log; cat $L.tf_other.log
B="python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timer"
run() { tag=$1; shift; timeout 300 env "$@" $B 2>$L.$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('final_loss'))" >> $L.ab.log 2>&1; }
: > $L.ab.log
run base X=1
run state_bf16 CRIS_STATE_FP32=0
run base2 X=1
echo "=== step A/B (sentence vector in fp32)"; cat $L.ab.log
B2="python bench.py --path module --steps 60 --warmup 10 --no-cpu-baseline --phase-times"
mp() { tag=$1; shift; timeout 300 env "$@" $B2 2>$L.module_$tag.err | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('module/$tag', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('optimizer'), d['config'].get('replay'))
for k,v in (d.get('phase_times') or {}).items(): print('   %-14s host %.2f ms  device %.2f ms' % (k, v['host_ms'], v['device_ms']))" >> $L.module.log 2>&1; }
: > $L.module.log
mp torch X=1 ; sed -i 's/--phase-times/--phase-times --optimizer cris/' /dev/null
B2="$B2 --optimizer cris"; mp cris X=1
mp cris_cmdlist CRIS_MODULE_REPLAY=cmdlist
echo "=== module path"; cat $L.module.log
