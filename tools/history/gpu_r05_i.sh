#!/bin/bash
# Round 5, call I: the module tests incl. the replaced-middle-parameter test; the per-shape tile-variant table with this round's kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r05i
F="Warning\|warn\|return float\|Consider using\|amdgpu.ids\|Gloo\|c10d"
python -c "import __graft_entry__ as g; g.build()" || exit 1
( time timeout 900 python -m pytest tests/test_module_gpu.py tests/test_module_surface.py -m gpu -q -x -p no:cacheprovider --durations=4 ) 2>&1 | grep -v "$F" | tail -14 | cut -c1-300 > $L.module.log; tail -10 $L.module.log
( time timeout 900 python tools/gemm_variants.py --min-m 1000 --rounds 3 --tsv $L.gemm_variants.tsv ) 2>&1 | grep -v "$F" | tail -70 | cut -c1-220 > $L.variants.log; tail -45 $L.variants.log
