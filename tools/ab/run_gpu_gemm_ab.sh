#!/bin/bash
# A/B timing of the 256-tile GEMM schedules at the encoder shapes (B=8: M=12000): usage run_gpu_gemm_ab.sh TAG
TAG=${1:-ab}
mkdir -p gpurun_out
[ -z "$SKIPTEST" ] && python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm" -x 2>&1 | tail -3
cat > /tmp/gemm_ab.py <<'P'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import helpers as Hh
from crisperwhisper_amd.engine import Engine
g, v, W, spec = Hh.tiny_setup()
e = Engine(spec, dtype="bf16", max_batch=4)
rng = np.random.default_rng(0)
for pp, ph8 in ((0, 0), (1, 0), (1, 1)):
    e.lib.cw_test_set_option(b"gemm_pp", pp)
    e.lib.cw_test_set_option(b"gemm_8ph", ph8)
    sys.stderr.write(f"== ping-pong {pp} 8-phase {ph8}\n")
    for (M, N, K, gelu) in [(12000, 3840, 1280, False), (12000, 1280, 1280, False), (12000, 5120, 1280, True), (12000, 1280, 5120, False)]:
        A = rng.standard_normal((M, K)).astype(np.float32)
        Wm = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        e.test_gemm(A, Wm, None, gelu)
e.close()
P
CW_TEST_GEMM_REPS=30 python /tmp/gemm_ab.py 2>&1 | grep "==\|cw_test_gemm" | tee gpurun_out/gemm_ab_$TAG.txt
