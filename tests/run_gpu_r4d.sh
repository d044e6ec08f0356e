#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 400 python tools/skinny_bench.py 40 64 > gpurun_out/r4d_sweep.txt 2>&1
grep -v "atomics" gpurun_out/r4d_sweep.txt
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -k "large_batch or beam" 2>&1 | tail -30 > gpurun_out/r4d_rows.log
tail -8 gpurun_out/r4d_rows.log
for sk in 0 1; do
  if [ $sk = 0 ]; then export CW_NO_SKINNY=1; else unset CW_NO_SKINNY; fi
  timeout 600 python bench.py --batch 64 --tokens 32 --steps 2 --warmup 1 --no-cpu-baseline --no-config3 --no-longform > gpurun_out/r4d_b64_sk$sk.json 2> gpurun_out/r4d_b64_sk$sk.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4d_b64_sk$sk.json").read().strip().splitlines()[-1])
print("skinny=$sk B=64 ms_per_step", d["ms_per_step"], "decode", d.get("stage_roofline",{}).get("decode_step"), "parity", d.get("parity",{}).get("ok"))
PY
done
