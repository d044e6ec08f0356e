#!/bin/bash
# round 5: persistent decoder-layer kernel (csrc/declayer.hip) -- bit-identity tests, same-box A/B of the headline bench over the
# K/V queue depth, kernel trace.  usage: run_gpu_r5_declayer.sh TAG "depths" [profile-depth]
TAG=${1:-r5b}; DEPTHS=${2:-"4 8 16 2"}; PD=${3:-4}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "persistent_decoder_layer" > gpurun_out/${TAG}_tests.log 2>&1
tail -3 gpurun_out/${TAG}_tests.log
B="--steps 3 --warmup 1 --no-cpu-baseline --no-config3 --no-longform"
CW_NO_DECLAYER=1 timeout 600 python bench.py $B > gpurun_out/${TAG}_bench_launches.json 2> gpurun_out/${TAG}_bench_launches.err
for d in $DEPTHS; do
  CW_DL_DEPTH=$d timeout 600 python bench.py $B > gpurun_out/${TAG}_bench_depth$d.json 2> gpurun_out/${TAG}_bench_depth$d.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 2), round(d.get("stage_roofline", {}).get("decode_step", {}).get("ms_per_step"), 4), d.get("parity", {}).get("ok"))
    except Exception as e:
        print(f, "ERR", e)
PY
[ "$PD" != "0" ] && CW_DL_DEPTH=$PD bash tools/ab/run_gpu_prof_args.sh ${TAG}_depth$PD --no-config3 | head -12
