#!/bin/bash
# round 5: fc1 + fc2 in one launch (csrc/declayer.hip: mlp_chain_kernel) -- bit-identity tests, same-box A/B of the headline bench
# (CW_NO_MLP_CHAIN=1 = two launches), per-launch table.  usage: run_gpu_r5_mlpchain.sh TAG
TAG=${1:-r5m}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "persistent_decoder_layer and mlp_chain" > gpurun_out/${TAG}_tests.log 2>&1
tail -5 gpurun_out/${TAG}_tests.log
B="--steps 3 --warmup 1 --no-cpu-baseline --no-config3 --no-longform"
for rep in 1 2; do
CW_NO_MLP_CHAIN=1 timeout 600 python bench.py $B > gpurun_out/${TAG}_bench_launches$rep.json 2> gpurun_out/${TAG}_bench_launches$rep.err
timeout 600 python bench.py $B > gpurun_out/${TAG}_bench_fused$rep.json 2> gpurun_out/${TAG}_bench_fused$rep.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 2), round(d["stage_roofline"]["decode_step"]["ms_per_step"], 4), d["parity"]["ok"])
        for r in [d["roofline"]] + d["roofline_other"]:
            print("   %-75s %7.2f us share %.3f" % (r["kernel"][:75], r["avg_launch_ms"] * 1e3, r["share_of_decode_step"] or 0))
    except Exception as e:
        print(f, "ERR", e)
PY
