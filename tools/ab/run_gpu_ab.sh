#!/bin/bash
# A/B of a compile-time switch: builds attention.hip with and without -D$1, runs microbench + short bench for each, twice interleaved
cd $GRAFT_REPO_ROOT/crisperwhisper_amd/csrc
for rep in 1 2; do
for flag in "" "-D$1"; do
  rm -f build/attention.o
  make -s CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $flag" > /dev/null 2>&1
  echo "== flag='$flag'"
  (cd ../.. && python tests/gpu_microbench.py 8 | grep -E "cross-attn|self-attn"; python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['stage_ms_per_step']['decode'])")
done
done
