#!/bin/bash
# round 2, call D: wide-store GEMM epilogue, grouped wgrads on a second stream (A/B), peer-mailbox SyncBN in the trainer
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02d
timeout 600 python -m pytest tests/test_hip_ops.py -q 2>&1 | tail -30 > $L.hip_ops.log
timeout 600 python -m pytest tests/test_engine_gpu.py -q -k "tiny_step or ragged or deterministic or r50_small" 2>&1 | tail -12 > $L.engine.log
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline"
timeout 300 $B --shape-table $L.shapes.tsv > $L.bench.json 2> $L.bench.err
CRIS_WGRAD_STREAM=1 timeout 300 $B --no-kernel-timer > $L.bench_wstream.json 2>> $L.bench.err
export CRIS_TEST_P2P=1
timeout 300 python -m pytest tests/test_p2p_gpu.py -q -x 2>&1 | tail -8 > $L.p2p.log
for p2p in 0 1; do
  echo "== bench 2 ranks on one GPU, CRIS_SYNCBN_P2P=$p2p" >> $L.p2p.log
  CRIS_SYNCBN_P2P=$p2p timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 2 --backend gloo --batch 4 --steps 10 --warmup 2 --no-kernel-timer 2>/dev/null | cut -c1-300 >> $L.p2p.log
done
for f in hip_ops engine p2p; do echo "=== $f"; tail -14 $L.$f.log; done
echo "=== bench"; cut -c1-420 $L.bench.json; echo; cut -c1-420 $L.bench_wstream.json; tail -3 $L.bench.err
