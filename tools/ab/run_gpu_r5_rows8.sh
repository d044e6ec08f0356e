#!/bin/bash
# round 5: beam search over the e4m3 cross-attention cache -- attn_cross_mfma8_rows_kernel (one block per (item, head, key split) for
# all hypotheses of the item) against one block per hypothesis row: kernel tests, the bench-model beam test, kernel time per launch
# at 8 items x 5 hypotheses, and `bench.py --num-beams 5 --cross-kv fp8` both ways on the same box.  usage: run_gpu_r5_rows8.sh TAG
TAG=${1:-r5rows8}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
rm -f gpurun_out/${TAG}_errlog.txt
CW_TEST_ERRLOG=$R/gpurun_out/${TAG}_errlog.txt timeout 420 python -m pytest tests/test_gpu_kernels.py -x -q -p no:cacheprovider \
    -k "e4m3_cache or test_gemm or test_gemv" > gpurun_out/${TAG}_ktests.log 2>&1
tail -4 gpurun_out/${TAG}_ktests.log
timeout 420 python -m pytest tests/test_gpu_e2e.py -x -q -s -p no:cacheprovider -k "e4m3_cache_rows_kernel" > gpurun_out/${TAG}_e2e.log 2>&1
grep -E "e4m3 cache, 5 beams|passed|failed|Error" gpurun_out/${TAG}_e2e.log | tail -5
CW_TEST_ATTN_REPS=300 timeout 200 python - > gpurun_out/${TAG}_kernel_us.txt 2>&1 <<'PY'
import numpy as np
from tests import helpers as Hh
from crisperwhisper_amd.engine import Engine
from crisperwhisper_amd import _native
g, v, W, spec = Hh.tiny_setup()
e = Engine(spec, dtype="bf16", max_batch=4)
lib = _native.load()
rng = np.random.default_rng(0)
for items, nq in ((8, 5), (8, 2), (8, 8)):
    B, H, S = items * nq, 20, 1500
    q = (rng.standard_normal((B, H, 64)) * 0.35).astype(np.float32)
    k = rng.standard_normal((items, H, S, 64)).astype(np.float32); vv = rng.standard_normal((items, H, S, 64)).astype(np.float32)
    for fp8 in (1, 0):
        lib.cw_test_set_option(b"cross_test_fp8", fp8)
        for per_row in (0, 1):
            lib.cw_test_set_option(b"cross_per_row", per_row)
            print(f"items={items} nq={nq} cache={'e4m3' if fp8 else 'bf16'} per_row={per_row}", flush=True)
            e.test_cross_attention(q, k, vv, kv_div=nq, align_head=0)
lib.cw_test_set_option(b"cross_test_fp8", 0); lib.cw_test_set_option(b"cross_per_row", 0)
e.close()
PY
cat gpurun_out/${TAG}_kernel_us.txt
run() {
  name=$1; shift
  env "$@" timeout 300 python bench.py --num-beams 5 --cross-kv fp8 --steps 2 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --no-rccl --kernel-iters 20 > gpurun_out/${TAG}_beam_$name.json 2> gpurun_out/${TAG}_beam_$name.err
  python - "$name" "$TAG" <<'P'
import json, sys
try:
    d = json.loads([l for l in open(f"gpurun_out/{sys.argv[2]}_beam_{sys.argv[1]}.json") if l.startswith("{")][-1])
    print(sys.argv[1], "ms_per_step", round(d["ms_per_step"], 1), "passes", d.get("passes_per_step"), "words/s", round(d["value"], 1),
          "decode ms per beam step", round(d["stage_roofline"]["decode_step"]["ms_per_step"], 3))
except Exception as ex:
    print(sys.argv[1], "ERR", ex)
P
}
run rows A=1
run per_row CW_CROSS_PER_ROW=1
python - "$TAG" <<'P'
import sys, collections
rows = collections.defaultdict(list)
try:
    for l in open(f"gpurun_out/{sys.argv[1]}_errlog.txt"):
        n, e = l.rstrip("\n").split("\t")
        base = n.split("::")[-1].split("[")[0]
        dt = "bf16" if "bf16" in n else "f16" if "f16" in n else "f32" if "f32" in n else "?"
        rows[(base, dt)].append(float(e))
    for k in sorted(rows):
        print(k, "n", len(rows[k]), "max", max(rows[k]))
except Exception as ex:
    print("errlog", ex)
P
