#!/bin/bash
# round 5: persistent decoder-layer kernel (csrc/declayer.hip) -- bit-identity tests, same-box A/B of the headline bench, kernel trace
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "persistent_decoder_layer or fused_decoder_stage or bit_reproducible" > gpurun_out/r5a_tests.log 2>&1
tail -5 gpurun_out/r5a_tests.log
for rep in 1 2; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config3 --no-longform > gpurun_out/r5a_bench_declayer_$rep.json 2> gpurun_out/r5a_bench_declayer_$rep.err
  CW_NO_DECLAYER=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config3 --no-longform > gpurun_out/r5a_bench_launches_$rep.json 2> gpurun_out/r5a_bench_launches_$rep.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5a_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d.get("stage_roofline", {}).get("decode_step", {}).get("ms_per_step"), d.get("parity", {}).get("ok"))
    except Exception as e:
        print(f, "ERR", e)
PY
bash tools/ab/run_gpu_prof_args.sh r5a_declayer --no-config3
