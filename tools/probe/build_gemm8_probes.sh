#!/bin/bash
# one probe binary per build spec "mask[:tag:extra -D flags]" (cross-compiles without a GPU; the binaries are git-ignored and
# travel with gpurun).   e.g.  build_gemm8_probes.sh 0 128 "0:ph8:-DG8_PH16=0" "0:direct:-DG8_LDS_EPI=0"
cd "$(dirname "$0")"
rm -f gemm8_probe_*
for spec in "$@"; do
  IFS=: read -r m tag extra <<< "$spec"
  out=gemm8_probe_$m${tag:+_$tag}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-vectorize -Wno-unused-result -Wno-unused-value -DG8_ABL=$m $extra gemm8_probe.hip -o $out 2>/dev/null &
done
wait
ls gemm8_probe_* | tr '\n' ' '
