#!/bin/bash
# round 5: fused out-projection / cross-query stage at 17..64 greedy rows (row groups of gemv_stack_kernel) against the twelve-launch
# layer (CW_NO_FUSE_ROWS=1): stage test, then the batch-64 bench both ways on one box (bf16 cache and the fp8 mode), every clip compared
# with its transformers reference.  usage: run_gpu_r5_fuserows.sh TAG [notests]
TAG=${1:-r5fuserows}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
if [ "$2" != "notests" ]; then
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -p no:cacheprovider -k "fused_decoder_stage_tracks" > gpurun_out/${TAG}_tests.log 2>&1
tail -5 gpurun_out/${TAG}_tests.log
fi
run() {
  name=$1; shift
  env "$@" timeout 400 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --no-rccl --kernel-iters 20 $EXTRA > gpurun_out/${TAG}_$name.json 2> gpurun_out/${TAG}_$name.err
  python - "$name" "$TAG" <<'P'
import json, sys
try:
    d = json.loads([l for l in open(f"gpurun_out/{sys.argv[2]}_{sys.argv[1]}.json") if l.startswith("{")][-1])
    print(sys.argv[1], "ms_per_step", round(d["ms_per_step"], 1), "decode ms per token step", round(d["stage_roofline"]["decode_step"]["ms_per_step"], 3), "parity", d.get("parity", {}).get("clips_with_identical_text"), d.get("parity", {}).get("words_identical_and_within_20ms"), "launches/layer", d["stage_roofline"]["decode_step"].get("launches_per_layer"))
except Exception as ex:
    print(sys.argv[1], "ERR", ex)
    print(open(f"gpurun_out/{sys.argv[2]}_{sys.argv[1]}.err").read()[-1500:])
P
}
EXTRA=""
run fused A=1
run twelve CW_NO_FUSE_ROWS=1
EXTRA="--cross-kv fp8"
run fused_fp8 A=1
run twelve_fp8 CW_NO_FUSE_ROWS=1
