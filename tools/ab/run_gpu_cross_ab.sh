#!/bin/bash
# cross-attention decode kernels: differential test + per-launch time at the beam-search shape (8 items x 5 hypotheses)
export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "cross_attention_decode" 2>&1 | tail -5
CW_CROSS_NO_TR=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "cross_attention_decode" 2>&1 | tail -3
cat > /tmp/xab.py <<'P'
import numpy as np, sys
from tests import helpers as Hh
from crisperwhisper_amd.engine import Engine
from crisperwhisper_amd import _native
g, v, W, spec = Hh.tiny_setup()
e = Engine(spec, dtype="bf16", max_batch=4)
rng = np.random.default_rng(0)
B, H, S, kd = 40, 20, 1500, 5
q = (rng.standard_normal((B, H, 64)) * 0.35).astype(np.float32)
k = rng.standard_normal((B // kd, H, S, 64)).astype(np.float32); vv = rng.standard_normal((B // kd, H, S, 64)).astype(np.float32)
lib = _native.load()
for path in (0, 1):
    lib.cw_test_set_option(b"cross_valu", path)
    sys.stderr.write("valu=%d " % path); sys.stderr.flush()
    e.test_cross_attention(q, k, vv, kv_div=kd, align_head=3)
P
PYTHONPATH=$GRAFT_REPO_ROOT CW_TEST_ATTN_REPS=200 timeout 200 python /tmp/xab.py 2>&1 | grep -E "us/launch|Error|error" 
CW_CROSS_NO_TR=1 PYTHONPATH=$GRAFT_REPO_ROOT CW_TEST_ATTN_REPS=200 timeout 200 python /tmp/xab.py 2>&1 | grep -E "us/launch|Error|error" | head -1
