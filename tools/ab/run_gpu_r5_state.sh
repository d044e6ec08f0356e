#!/bin/bash
# round 5 re-entry: where the tree stands -- persistent-layer tests, same-box A/B of the headline bench, kernel trace.
TAG=${1:-r5s}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "persistent_decoder_layer" > gpurun_out/${TAG}_tests.log 2>&1
tail -3 gpurun_out/${TAG}_tests.log
B="--steps 3 --warmup 1 --no-cpu-baseline --no-config3 --no-longform"
CW_NO_QKV_SELF=1 timeout 600 python bench.py $B > gpurun_out/${TAG}_bench_launches.json 2> gpurun_out/${TAG}_bench_launches.err
timeout 600 python bench.py $B > gpurun_out/${TAG}_bench_fused.json 2> gpurun_out/${TAG}_bench_fused.err
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 2), round(d.get("stage_roofline", {}).get("decode_step", {}).get("ms_per_step"), 4), d.get("parity", {}).get("ok"))
    except Exception as e:
        print(f, "ERR", e)
PY
bash tools/ab/run_gpu_prof_args.sh ${TAG}_fused --no-config3 | head -30
