#!/bin/bash
# Workload sweep of bench.py (other operating points of DESIGN 6a): usage run_gpu_sweep.sh TAG
TAG=${1:-sweep}
mkdir -p gpurun_out
for t in 64 224; do
  python bench.py --tokens $t --steps 3 --warmup 1 --no-cpu-baseline --no-longform 2>/dev/null | tail -1 > gpurun_out/${TAG}_T$t.json
done
for b in 1 16 32 64; do
  timeout 600 python bench.py --batch $b --steps 2 --warmup 1 --no-cpu-baseline --no-longform 2>gpurun_out/${TAG}_B$b.err | tail -1 > gpurun_out/${TAG}_B$b.json
done
timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-longform --cross-kv fp8 2>/dev/null | tail -1 > gpurun_out/${TAG}_B64_fp8kv.json
timeout 600 python bench.py --batch 16 --steps 2 --warmup 1 --no-cpu-baseline --no-longform --dtype f16 2>/dev/null | tail -1 > gpurun_out/${TAG}_B16_f16.json
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_*.json")):
    try:
        j = json.loads(open(f).read())
        print(f, round(j["ms_per_step"], 1), "ms  rtf", round(j["rtf"], 5), " words/s", round(j["value"], 1), j["stage_ms_per_step"], j.get("parity", {}) and j["parity"].get("clips_with_identical_text"))
    except Exception as e:
        print(f, "FAILED", e)
PY
