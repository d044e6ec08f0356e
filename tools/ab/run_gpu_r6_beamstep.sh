#!/bin/bash
# beam-search decoder step, one controlled pass (tools/beam_step_bench.py): twelve launches / fused stage / fused stage + column-owning out-projection
for e in "CW_NO_FUSE_BEAM=1" "CW_NO_OWN_COLS=1" "CW_OWN_NT=1" "CW_OWN_NT=2" "CW_NO_FUSE_BEAM=1" "CW_NO_OWN_COLS=1" "CW_OWN_NT=1"; do
  echo "== $e"; env $e python tools/beam_step_bench.py 2>&1 | tail -1
done
