#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 -p no:cacheprovider -k "mel" 2>&1 | tail -8
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config3 --no-longform > gpurun_out/r4j_bench.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/r4j_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["stage_ms_per_step"], d["stage_roofline"]["mel"]["ms_per_call"], d["parity"]["ok"])
PY
