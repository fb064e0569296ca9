#!/bin/bash
# one probe binary per build spec "tag[:extra -D flags]" (cross-compiles without a GPU; binaries are git-ignored and travel with gpurun)
#   build_gemm4_probes.sh full "nomfma:-DG4_ABL=1" "nodma:-DG4_ABL=2" "noepi:-DG4_ABL=4" "st5:-DST_64x64=5"
cd "$(dirname "$0")"
rm -f gemm4_probe_*
for spec in "$@"; do
  IFS=: read -r tag extra <<< "$spec"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-vectorize -Wno-unused-result -Wno-unused-value -DG4_PROBE $extra gemm4_probe.hip -o gemm4_probe_$tag 2>gemm4_build_$tag.log &
done
wait
grep -l "error" gemm4_build_*.log 2>/dev/null | head; rm -f gemm4_build_*.log
ls gemm4_probe_* | tr '\n' ' '
