#!/bin/bash
# round 5, end of round: the default bench line (every leg, CPU reference included), kernel tables of the headline configuration and
# of batch 64 under rocprofv3.  usage: run_gpu_r5_final.sh TAG
TAG=${1:-r5final}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 1200 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["rtf"], d["parity"]["ok"], d["roofline"]["frac"], d["roofline"]["step_frac"])
print(d["cpu_baseline"])
print({k: (round(v["ms_per_step"], 1), v.get("parity_ok")) for k, v in d["config3"]["modes"].items()})
print(d["longform"]["wall_s"], d["longform"]["words"])
PY
bash tools/ab/run_gpu_prof_args.sh ${TAG}_B8 --no-config3 | head -14
bash tools/ab/run_gpu_prof_args.sh ${TAG}_B64 --no-config3 --batch 64 | head -18
