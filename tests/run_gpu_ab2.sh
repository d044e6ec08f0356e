#!/bin/bash
# A/B of a compile-time switch in gemm.hip: microbench both builds, twice interleaved
cd $GRAFT_REPO_ROOT/crisperwhisper_amd/csrc
for rep in 1 2; do
for flag in "" "-D$1"; do
  rm -f build/gemm.o
  make -s CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $flag" > /dev/null 2>&1
  echo "== flag='$flag'"
  (cd ../.. && python tests/gpu_microbench.py 8 | grep -E "fc1|qkv|q_c|fc2|o-proj")
done
done
