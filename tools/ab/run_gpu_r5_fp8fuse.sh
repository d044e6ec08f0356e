#!/bin/bash
# round 5: e4m3 cross-attention cache composed with the fused out-projection / cross-query stage -- tests, then the B = 8 bench in
# that mode against the 8-launch layer (CW_NO_FUSE6=1) on the same box.
TAG=${1:-r5f8}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_kernels.py -x -q -m gpu -k "fused_decoder_stage_tracks or fp8_cross_kv or e4m3" > gpurun_out/${TAG}_tests.log 2>&1
tail -3 gpurun_out/${TAG}_tests.log
B="--steps 3 --warmup 1 --no-cpu-baseline --no-config3 --no-longform --cross-kv fp8"
CW_NO_FUSE6=1 timeout 600 python bench.py $B > gpurun_out/${TAG}_bench_eight.json 2> gpurun_out/${TAG}_bench_eight.err
timeout 600 python bench.py $B > gpurun_out/${TAG}_bench_fused.json 2> gpurun_out/${TAG}_bench_fused.err
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 2), round(d["stage_roofline"]["decode_step"]["ms_per_step"], 4), d["parity"]["ok"], d["parity"]["clips_with_identical_text"], d["parity"]["words_identical_and_within_20ms"])
        print("   ", d["roofline"]["kernel"], round(d["roofline"]["avg_launch_ms"] * 1e3, 2), "us", round(d["roofline"]["frac"], 3))
    except Exception as e:
        print(f, "ERR", e)
PY
