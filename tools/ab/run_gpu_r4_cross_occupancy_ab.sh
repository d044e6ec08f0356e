#!/bin/bash
# cross-attention at B = 8: blocks per CU limited by an LDS pad (CW_CROSS_LDS_PAD): does staggering the blocks' tails pay?
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for pad in 0 36000 50000 75000 140000; do
  export CW_CROSS_LDS_PAD=$pad
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config3 --no-longform > gpurun_out/r4x_b.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4x_b.json").read().strip().splitlines()[-1])
print("lds pad $pad: step", round(d["ms_per_step"],1), "decode/step", round(d["stage_roofline"]["decode_step"]["ms_per_step"],4), "cross-attn us", round(d["roofline"]["avg_launch_ms"]*1e3,2), "frac", round(d["roofline"]["frac"],4), "parity", d["parity"]["ok"])
PY
done
