#!/bin/bash
# workload sweep around the headline config: tokens/chunk 64/128/224 and batch 16/32/64 (configs[4] shape, bf16)
mkdir -p gpurun_out
for t in 64 224; do
  python bench.py --tokens $t --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/sweep_T$t.json
done
for b in 16 32 64; do
  timeout 600 python bench.py --batch $b --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/sweep_B$b.err | tail -1 > gpurun_out/sweep_B$b.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/sweep_*.json")):
    try:
        j = json.loads(open(f).read())
        print(f, round(j["ms_per_step"], 1), "ms  rtf", round(j["rtf"], 5), " words/s", round(j["value"], 1), " tok/s", round(j["tokens_per_s"], 1), j["stage_ms_per_step"])
    except Exception as e:
        print(f, "FAILED", e)
PY
