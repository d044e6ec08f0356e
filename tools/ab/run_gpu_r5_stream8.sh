#!/bin/bash
# round 5: e4m3 cache at 17..64 greedy rows -- attn_cross_mfma8_stream_kernel (persistent blocks, next item's bytes in flight) against
# one block per item: bit-identity tests, us per launch at 64 / 40 / 20 rows, the batch-64 bench in the fp8 mode both ways.
TAG=${1:-r5stream8}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -p no:cacheprovider -k "e4m3_cache" > gpurun_out/${TAG}_ktests.log 2>&1
tail -4 gpurun_out/${TAG}_ktests.log
CW_TEST_ATTN_REPS=300 timeout 200 python - > gpurun_out/${TAG}_kernel_us.txt 2>&1 <<'PY'
import numpy as np
from tests import helpers as Hh
from crisperwhisper_amd.engine import Engine
from crisperwhisper_amd import _native
g, v, W, spec = Hh.tiny_setup()
e = Engine(spec, dtype="bf16", max_batch=4)
lib = _native.load()
rng = np.random.default_rng(0)
lib.cw_test_set_option(b"cross_test_fp8", 1)
for B in (64, 40, 20):
    H, S = 20, 1500
    q = (rng.standard_normal((B, H, 64)) * 0.35).astype(np.float32)
    k = rng.standard_normal((B, H, S, 64)).astype(np.float32); vv = rng.standard_normal((B, H, S, 64)).astype(np.float32)
    for stream in (1, 0, 1, 0):
        lib.cw_test_set_option(b"cross8_stream", stream)
        print(f"B={B} stream={stream}", flush=True)
        e.test_cross_attention(q, k, vv, kv_div=1, align_head=0)
lib.cw_test_set_option(b"cross_test_fp8", 0); lib.cw_test_set_option(b"cross8_stream", 1)
e.close()
PY
paste -d' ' - - < gpurun_out/${TAG}_kernel_us.txt
if [ "$2" != "nobench" ]; then
run() {
  name=$1; shift
  env "$@" timeout 400 python bench.py --batch 64 --cross-kv fp8 --steps 2 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --no-rccl --kernel-iters 20 > gpurun_out/${TAG}_b64_$name.json 2> gpurun_out/${TAG}_b64_$name.err
  python - "$name" "$TAG" <<'P'
import json, sys
try:
    d = json.loads([l for l in open(f"gpurun_out/{sys.argv[2]}_b64_{sys.argv[1]}.json") if l.startswith("{")][-1])
    print(sys.argv[1], "ms_per_step", round(d["ms_per_step"], 1), "decode ms per token step", round(d["stage_roofline"]["decode_step"]["ms_per_step"], 3), "parity", d.get("parity", {}).get("clips_with_identical_text"), d.get("parity", {}).get("words_identical_and_within_20ms"))
except Exception as ex:
    print(sys.argv[1], "ERR", ex)
P
}
run stream A=1
run one_per_item CW_CROSS8_NO_STREAM=1
fi
