#!/bin/bash
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "teacher_forced or bf16_runs or large_geometry_layers or full_size_bf16 or bench_shape or reproducible" 2>&1 | tail -8
bash tests/run_gpu_prof2.sh pair A=1 -- > /dev/null 2>&1
grep -E "cross_|gemv_stack|mlp_pair|gemv2_bf16|attn_decode" gpurun_out/prof_pair.txt | cut -c1-75,100-160
ARGS="--batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --no-rccl --kernel-iters 50"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $ARGS > gpurun_out/r3c_$name.json 2> gpurun_out/r3c_$name.err;
  python - <<P
import json
try:
    d=json.load(open("gpurun_out/r3c_$name.json")); print("$name", round(d["ms_per_step"],1), d["stage_ms_per_step"], round(d["stage_roofline"]["decode_step"]["ms_per_step"],4), d["parity"]["ok"])
except Exception as e: print("$name failed", e); print(open("gpurun_out/r3c_$name.err").read()[-800:])
P
}
run pair A=1
run nopair CW_NO_MLP_PAIR=1
