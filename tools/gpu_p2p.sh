#!/bin/bash
# First GPU call for the experimental peer-mailbox SyncBN exchange (csrc/p2p.hip, dist.PeerMailboxes; CRIS_SYNCBN_P2P=1):
# two ranks on the one GPU of the box, primitive first (short timeout: it spins on the peer), then the trainer comparison,
# then the 2-rank bench step time with and without it (gloo carries the gradient exchange in both, so only the difference
# between the two lines means anything).
#   gpurun --timeout 500 -- 'bash tools/gpu_p2p.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; out=gpurun_out/p2p.log; : > $out
export CRIS_TEST_P2P=1
timeout 120 python -m pytest tests/test_p2p_gpu.py -q -x -k peer_mailbox 2>&1 | tail -5 >> $out
timeout 300 python -m pytest tests/test_p2p_gpu.py -q -x -k trainer 2>&1 | tail -5 >> $out
for p2p in 0 1; do
  echo "== bench 2 ranks on one GPU, CRIS_SYNCBN_P2P=$p2p" >> $out
  CRIS_SYNCBN_P2P=$p2p timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 2 --backend gloo --batch 4 --steps 10 --warmup 2 --no-kernel-timer 2>/dev/null | cut -c1-300 >> $out
done
cat $out
