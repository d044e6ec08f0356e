#!/bin/bash
# round 5: the measured inputs of DESIGN.md section 5's scaling prediction -- the per-pass stage times at 4 / 6 / 7 / 8 chunks per
# rank (what a rank of the 2 / 4 / 8-GPU long-form run executes), the gather cost through RCCL at world size 1, the long-form wall.
TAG=${1:-r5sc}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for b in 4 6 7; do
  timeout 600 python bench.py --batch $b --steps 2 --warmup 1 --no-cpu-baseline --no-config3 --no-longform > gpurun_out/${TAG}_b$b.json 2> gpurun_out/${TAG}_b$b.err
done
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config3 > gpurun_out/${TAG}_b8.json 2> gpurun_out/${TAG}_b8.err
python - <<PY
import json
for b in (4, 6, 7, 8):
    d = json.loads(open("gpurun_out/${TAG}_b%d.json" % b).read().strip().splitlines()[-1])
    sr = d["stage_roofline"]
    print(b, round(d["ms_per_step"], 1), {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}, "dec step", round(sr["decode_step"]["ms_per_step"], 4), d["collective"]["gather_us_per_call"], d["collective"]["all_gathers_in_timed_region"])
    if d.get("longform"):
        print("   longform", {k: d["longform"][k] for k in ("wall_s", "rtf", "words", "chunk_shards")})
PY
