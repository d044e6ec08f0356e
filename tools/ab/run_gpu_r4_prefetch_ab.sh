#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
run() {
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config3 --no-longform > gpurun_out/r4n_b.json 2>gpurun_out/r4n_b.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r4n_b.json").read().strip().splitlines()[-1])
    print("$1 step", round(d["ms_per_step"],1), "decode ms/step", round(d["stage_roofline"]["decode_step"]["ms_per_step"],4), "parity", d["parity"]["ok"])
except Exception as e:
    print("$1 failed", e); print(open("gpurun_out/r4n_b.err").read()[-800:])
PY
}
export CW_PREFETCH=0; run "off"
export CW_PREFETCH=256 CW_PREFETCH_WIDE=1 CW_PREFETCH_WHAT=3; run "graph wide all 256"
export CW_PREFETCH=256 CW_PREFETCH_WIDE=1 CW_PREFETCH_WHAT=1; run "graph wide weights 256"
export CW_PREFETCH=256 CW_PREFETCH_WIDE=0 CW_PREFETCH_WHAT=3; run "graph sector all 256"
export CW_NO_GRAPH=1
export CW_PREFETCH=0; run "eager off"
export CW_PREFETCH=256 CW_PREFETCH_WIDE=1 CW_PREFETCH_WHAT=3; run "eager wide all 256"
export CW_PREFETCH=256 CW_PREFETCH_WIDE=0 CW_PREFETCH_WHAT=3; run "eager sector all 256"
export CW_PREFETCH=256 CW_PREFETCH_WIDE=0 CW_PREFETCH_WHAT=1; run "eager sector weights 256"
export CW_PREFETCH=64 CW_PREFETCH_WIDE=1 CW_PREFETCH_WHAT=3; run "eager wide all 64"
