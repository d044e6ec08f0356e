#!/bin/bash
# e4m3 GEMM: differential tests + timing at the encoder shapes next to the bf16 ping-pong kernel: usage run_gpu_gemm_fp8.sh TAG
TAG=${1:-f8}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "fp8" -x 2>&1 | tail -5
cat > /tmp/gemm_f8.py <<'P'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import helpers as Hh
from crisperwhisper_amd.engine import Engine
g, v, W, spec = Hh.tiny_setup()
e = Engine(spec, dtype="bf16", max_batch=4)
rng = np.random.default_rng(0)
for (M, N, K, gelu) in [(12000, 3840, 1280, False), (12000, 5120, 1280, True), (12000, 1280, 5120, False), (96000, 5120, 1280, True)]:
    A = rng.standard_normal((M, K)).astype(np.float32)
    Wm = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    e.test_gemm(A, Wm, None, gelu)
    e.test_gemm_fp8(A, Wm, None, gelu)
e.close()
P
CW_TEST_GEMM_REPS=20 python /tmp/gemm_f8.py 2>&1 | grep "cw_test_gemm" | tee gpurun_out/gemm_fp8_$TAG.txt
