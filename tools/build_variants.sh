#!/bin/bash
# A/B builds of libcris_hip.so: the same sources with extra -D switches, cross-compiled here (no GPU needed) into
# cris/pytorch_amd/csrc/variants/libcris_hip_<tag>.so (git-ignored, travels with gpurun); CRIS_LIB_VARIANT=<tag> loads one (hip.py).
#   tools/build_variants.sh "sc1:-DCRIS_STORE_POLICY=1" "nt:-DCRIS_STORE_POLICY=2"
set -e
cd "$(dirname "$0")/../cris/pytorch_amd/csrc"
mkdir -p variants
SRC="api.hip gemm.hip gemm8.hip wgrad.hip norm.hip attention.hip elementwise.hip smallf32.hip evalpost.hip inputpipe.hip p2p.hip comm.hip jpeg.hip png.hip"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -fno-slp-vectorize -fno-vectorize"
for spec in "$@"; do
  tag="${spec%%:*}"; extra="${spec#*:}"
  d=variants/obj_$tag; mkdir -p $d
  for f in $SRC; do
    /opt/rocm/bin/hipcc $FLAGS $extra -c $f -o $d/${f%.hip}.o 2>/dev/null &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libcris_hip_$tag.so $d/*.o
  rm -rf $d
  echo "built variants/libcris_hip_$tag.so ($extra)"
done
