#!/bin/bash
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_e2e.py tests/test_dist_gloo.py -x -q -m gpu -k "second_weight_seed or unknown_generate or gpus2 or bench_shape or native_seek or beam_search_word" 2>&1 | tail -15
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-longform > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; tail -c 600 gpurun_out/r3a_bench.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r3a_bench.json"))
print(d["ms_per_step"], d["parity"]["ok"], d["stage_ms_per_step"])
print(json.dumps(d.get("config3"), indent=0)[:2500])
P
