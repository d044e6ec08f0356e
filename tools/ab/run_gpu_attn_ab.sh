#!/bin/bash
# A/B timing of the encoder attention kernels at the bench shape (B=8, 20 heads, S=1500): usage run_gpu_attn_ab.sh TAG
TAG=${1:-ab}
mkdir -p gpurun_out
[ -z "$SKIPTEST" ] && python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "encoder_attention" -x 2>&1 | tail -3
cat > /tmp/attn_ab.py <<'P'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import helpers as Hh
from crisperwhisper_amd.engine import Engine
g, v, W, spec = Hh.tiny_setup()
for dt in ("bf16", "f16"):
    e = Engine(spec, dtype=dt, max_batch=4)
    rng = np.random.default_rng(0)
    B, H, S = 8, 20, 1500
    q = (rng.standard_normal((B, H, S, 64)) * 0.3).astype(np.float32)
    k = rng.standard_normal((B, H, S, 64)).astype(np.float32)
    vv = rng.standard_normal((B, H, S, 64)).astype(np.float32)
    sys.stderr.write(f"dtype {dt} variant {'v1' if os.environ.get('CW_ATTN_V1') else 'q64'}\n")
    e.test_attention(q, k, vv)
    e.close()
P
for v in v1 q64; do
  unset CW_ATTN_V1
  if [ $v = v1 ]; then export CW_ATTN_V1=1; fi
  echo "== $v"
  CW_TEST_ATTN_REPS=100 python /tmp/attn_ab.py 2>&1 | grep -v "^\[W\|warn" | grep "B=8"
done | tee gpurun_out/attn_ab_$TAG.txt
