#!/bin/bash
# round 5: full default bench line (all legs except the CPU reference) with the per-launch roofline table.
# usage: run_gpu_r5_bench.sh TAG
TAG=${1:-r5p}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["parity"]["ok"], json.dumps(d["roofline"], indent=0)[:1500])
for r in d["roofline_other"]:
    print("%-75s %7.2f us  %7.1f GB/s  share %.3f" % (r["kernel"][:75], r["avg_launch_ms"] * 1e3, r["achieved_GBps"], r["share_of_decode_step"] or 0))
print(json.dumps(d["stage_roofline"]["decode_step"]))
print({k: (v.get("ms_per_step"), v.get("parity_ok")) for k, v in (d.get("config3") or {}).get("modes", {}).items()})
PY
