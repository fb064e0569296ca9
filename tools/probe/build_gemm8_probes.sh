#!/bin/bash
# one probe binary per ablation mask (cross-compiles without a GPU; the binaries are git-ignored and travel with gpurun)
cd "$(dirname "$0")"
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-vectorize -Wno-unused-result -Wno-unused-value -DG8_ABL=$m gemm8_probe.hip -o gemm8_probe_$m &
done
wait
ls -la gemm8_probe_* | head -30
