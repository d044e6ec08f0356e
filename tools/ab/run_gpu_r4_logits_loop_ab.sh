#!/bin/bash
# logits projection: persistent column loop (gemv_loop_kernel) against three tiles per block; bit-identity test + step timing
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --timeout=600 -p no:cacheprovider -k "persistent_column_loop" 2>&1 | tail -4
for nl in 0 1 0 1; do
  if [ $nl = 1 ]; then export CW_NO_GEMV_LOOP=1; else unset CW_NO_GEMV_LOOP; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config3 --no-longform > gpurun_out/r4y_b.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4y_b.json").read().strip().splitlines()[-1])
print("no_loop=$nl step", round(d["ms_per_step"],2), "decode/step", round(d["stage_roofline"]["decode_step"]["ms_per_step"],4), "parity", d["parity"]["ok"])
PY
done
