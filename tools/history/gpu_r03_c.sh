#!/bin/bash
# round 3, call C: 16-MFMA phases + LDS-staged epilogue of the 8-wave tiles: kernel tests, probes, per-shape table
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03c
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 400 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "8wave" 2>&1 | grep -v "$F" | tail -15 | cut -c1-400 > $L.kernel_tests.log
echo "=== kernel tests"; cat $L.kernel_tests.log
: > $L.probe.log
for shape in "0 8 104 256 512 3" "0 8 52 512 512 3" "2 8 104 512 256 3" "2 8 52 512 256 3" "1 8 104 128 128 3" "0 8 26 512 1024 3"; do
  v=${shape%% *}
  for b in tools/probe/gemm8_probe_*; do
    case $b in *ph8*) [ "$v" != "0" ] && continue;; esac
    echo -n "$(basename $b) " >> $L.probe.log
    timeout 60 $b $shape 20 2>&1 | grep G8PROBE >> $L.probe.log || echo >> $L.probe.log
  done
done
echo "=== probes"; cat $L.probe.log
timeout 400 python tools/gemm_variants.py --min-m 5000 --min-k 256 --rounds 3 --variants 128x128,64x128,64x64,8w256x256,8w256x128,8w128x256 --tsv $L.variants.tsv 2>&1 | grep "GEMMVAR\|Error\|error" | cut -c1-300 > $L.variants.log
echo "=== variants"; cat $L.variants.log
