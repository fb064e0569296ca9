#!/bin/bash
# round 2, call A: first GPU run of the transposing-read wgrad kernel, full-size parity figures, peer-mailbox primitive
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
bash tools/gpu_wgrad_tr.sh > /dev/null 2>&1
out=gpurun_out/r02a_parity.log; : > $out
for cfg in "r50 8 416 None" "r101 8 416 None" "r50 8 480 22"; do
  set -- $cfg
  echo "== selfcheck $cfg" >> $out
  timeout 400 python -c "
from cris.pytorch_amd import selfcheck
import json
print(json.dumps(selfcheck.run('$1', batch=$2, size=$3, dropout=0.0, word_len=$4)))" 2>&1 | tail -3 >> $out
done
export CRIS_TEST_P2P=1
timeout 120 python -m pytest tests/test_p2p_gpu.py -q -x -k peer_mailbox 2>&1 | tail -5 > gpurun_out/r02a_p2p.log
cat gpurun_out/wgrad_tr.log $out gpurun_out/r02a_p2p.log
