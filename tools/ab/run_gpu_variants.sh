#!/bin/bash
# builds attention.hip/gemm.hip/engine.hip variants with different split settings and microbenchmarks them
cd $GRAFT_REPO_ROOT/crisperwhisper_amd/csrc
for cfg in "4 512" "8 256" "8 512" "4 256" "6 512" "16 256" "10 256" "16 128"; do
  set -- $cfg
  rm -f build/attention.o build/gemm.o build/engine.o
  make -s -j8 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -DATT_NS=$1 -DCROSS_THREADS=$2" > /dev/null 2>&1
  echo "ATT_NS=$1 CROSS_THREADS=$2"
  (cd ../.. && python tests/gpu_microbench.py 8 | grep -E "cross-attn")
done
