#!/bin/bash
# round 5: fused stage at 17..64 rows over the e4m3 cache, the cross-attention kernel finishing the query in wave 0 alone
# (attn_cross_mfma8_kernel<1, 2>), against the twelve-launch layer (CW_NO_FUSE_ROWS8=1): stage test (fp8 rows), batch-64 fp8 bench both ways.
TAG=${1:-r5fuserows8}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
timeout 300 python -m pytest tests/test_gpu_e2e.py -x -q -p no:cacheprovider -k "fused_decoder_stage_tracks and fp8" > gpurun_out/${TAG}_tests.log 2>&1
tail -3 gpurun_out/${TAG}_tests.log
run() {
  name=$1; shift
  env "$@" timeout 300 python bench.py --batch 64 --cross-kv fp8 --steps 2 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --no-rccl --kernel-iters 5 > gpurun_out/${TAG}_$name.json 2> gpurun_out/${TAG}_$name.err
  python - "$name" "$TAG" <<'P'
import json, sys
try:
    d = json.loads([l for l in open(f"gpurun_out/{sys.argv[2]}_{sys.argv[1]}.json") if l.startswith("{")][-1])
    print(sys.argv[1], "ms_per_step", round(d["ms_per_step"], 1), "decode ms per token step", round(d["stage_roofline"]["decode_step"]["ms_per_step"], 3), "parity", d.get("parity", {}).get("clips_with_identical_text"), d.get("parity", {}).get("words_identical_and_within_20ms"))
    for r in d.get("roofline_other", []) + [d["roofline"]]:
        if "cross-attention" == r["kernel"] or "stack" in r["kernel"]: print("    ", r["kernel"][:70], round(r["avg_launch_ms"] * 1e3, 2), "us")
except Exception as ex:
    print(sys.argv[1], "ERR", ex)
P
}
run fused8 A=1
run twelve8 CW_NO_FUSE_ROWS8=1
