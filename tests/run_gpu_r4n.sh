#!/bin/bash
# run-ahead prefetch A/B (CW_PREFETCH = blocks of the side-stream launch; 0 = off)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for pf in 0 32 64 128 256 0; do
  export CW_PREFETCH=$pf
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config3 --no-longform > gpurun_out/r4n_bench_pf$pf.json 2>gpurun_out/r4n_pf$pf.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r4n_bench_pf$pf.json").read().strip().splitlines()[-1])
    print("prefetch=$pf step", round(d["ms_per_step"],1), "decode ms/step", round(d["stage_roofline"]["decode_step"]["ms_per_step"],4), "parity", d["parity"]["ok"])
except Exception as e:
    print("prefetch=$pf failed", e); print(open("gpurun_out/r4n_pf$pf.err").read()[-800:])
PY
done
