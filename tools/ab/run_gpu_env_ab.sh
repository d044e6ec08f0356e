#!/bin/bash
# A/B of HIP runtime settings that could move the per-launch floor of the decode chain (one bench run per setting).
export PYTHONUNBUFFERED=1
ARGS="--batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --no-rccl --kernel-iters 50"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $ARGS > gpurun_out/env_$name.json 2> gpurun_out/env_$name.err; 
  python - <<P
import json
try:
    d=json.load(open("gpurun_out/env_$name.json")); print("$name", round(d["ms_per_step"],1), d["stage_ms_per_step"], round(d["stage_roofline"]["decode_step"]["ms_per_step"],4), d["parity"]["clips_with_identical_text"], d["parity"]["words_identical_and_within_20ms"])
except Exception as e: print("$name failed", e)
P
}
run base A=1
run base2 A=1
run optflush0 AMD_OPT_FLUSH=0
run optflush1 AMD_OPT_FLUSH=1
run nograph CW_NO_GRAPH=1
run nofuse CW_NO_FUSE6=1
run directdisp0 AMD_DIRECT_DISPATCH=0
