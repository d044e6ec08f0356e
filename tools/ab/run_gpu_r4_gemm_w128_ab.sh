#!/bin/bash
# round 4: the four-wave 128 x 128 GEMM (csrc/gemm_w128.hip) against the 8-phase schedule: bit-identity tests, kernel timing at
# the encoder shapes, whole step
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm" -p no:cacheprovider 2>&1 | tail -4
cat > /tmp/gemm_ab.py <<'P'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import helpers as Hh
from crisperwhisper_amd.engine import Engine
g, v, W, spec = Hh.tiny_setup()
e = Engine(spec, dtype="bf16", max_batch=4)
rng = np.random.default_rng(0)
for w128 in (0, 1):
    e.lib.cw_test_set_option(b"gemm_w128", w128)
    sys.stderr.write(f"== w128 {w128}\n")
    for (M, N, K, gelu) in [(12000, 3840, 1280, False), (12000, 1280, 1280, False), (12000, 5120, 1280, True), (12000, 1280, 5120, False), (96000, 5120, 1280, True), (96000, 1280, 5120, False)]:
        A = rng.standard_normal((M, K)).astype(np.float32)
        Wm = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        e.test_gemm(A, Wm, None, gelu)
e.close()
P
CW_TEST_GEMM_REPS=20 timeout 900 python /tmp/gemm_ab.py 2>&1 | grep "==\|cw_test_gemm" | tee gpurun_out/r4p_gemm_ab.txt
for w in 1 0; do
  if [ $w = 0 ]; then export CW_NO_GEMM_W128=1; else unset CW_NO_GEMM_W128; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config3 --no-longform > gpurun_out/r4p_b.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4p_b.json").read().strip().splitlines()[-1])
print("w128=$w step", round(d["ms_per_step"],1), d["stage_ms_per_step"], "enc frac", round(d["stage_roofline"]["encoder"]["frac_of_2500TFps"],4), "ckv frac", round(d["stage_roofline"]["cross_kv"]["frac_of_2500TFps"],4), "parity", d["parity"]["ok"])
PY
done
