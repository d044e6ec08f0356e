#!/bin/bash
# round 5: stage A of declayer.hip with the K/V stream held back until the chain's tile has landed (CW_DL_KVWAIT) -- phase stamps
# (libcw_phase.so = make BUILD=build_ph OUT=../libcw_phase.so EXTRA=-DCW_PHASE_TIMING) and the headline bench.
TAG=${1:-r5k}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for w in 0 1 2; do
  echo "=== CW_DL_KVWAIT=$w" >> gpurun_out/${TAG}_phases.txt
  CW_LIB_PATH=$R/crisperwhisper_amd/libcw_phase.so CW_DL_KVWAIT=$w timeout 300 python tools/dl_phase_probe.py 8 >> gpurun_out/${TAG}_phases.txt 2>&1
done
cat gpurun_out/${TAG}_phases.txt
B="--steps 3 --warmup 1 --no-cpu-baseline --no-config3 --no-longform"
for w in 0 1 2; do
  CW_DECLAYER=1 CW_DL_KVWAIT=$w timeout 600 python bench.py $B > gpurun_out/${TAG}_bench_dl_wait$w.json 2> gpurun_out/${TAG}_bench_dl_wait$w.err
done
timeout 600 python bench.py $B > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 2), round(d.get("stage_roofline", {}).get("decode_step", {}).get("ms_per_step"), 4), d.get("parity", {}).get("ok"))
    except Exception as e:
        print(f, "ERR", e)
PY
