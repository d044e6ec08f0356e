#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for cfg in "8 1" "4 2" "2 4" "8 2"; do
  set -- $cfg
  timeout 600 python bench.py --batch $1 --contexts $2 --steps 3 --warmup 1 --no-cpu-baseline --no-config3 --no-longform > gpurun_out/r4o_b.json 2>gpurun_out/r4o_b.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r4o_b.json").read().strip().splitlines()[-1])
    print("batch $1 x contexts $2: step", round(d["ms_per_step"],1), "value", round(d["value"],1), "rtf", d["rtf"], "stages", d["stage_ms_per_step"], "parity", (d.get("parity") or {}).get("clips_with_identical_text"))
except Exception as e:
    print("failed", e); print(open("gpurun_out/r4o_b.err").read()[-600:])
PY
done
