#!/bin/bash
export PYTHONUNBUFFERED=1
ARGS="--batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --no-rccl --kernel-iters 50"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $ARGS > gpurun_out/r3b_$name.json 2> gpurun_out/r3b_$name.err;
  python - <<P
import json
try:
    d=json.load(open("gpurun_out/r3b_$name.json")); print("$name", round(d["ms_per_step"],1), d["stage_ms_per_step"], round(d["stage_roofline"]["encoder"]["frac_of_2500TFps"],4), d["parity"]["ok"])
except Exception as e: print("$name failed", e)
P
}
run gemm8ph A=1
run gemmpp CW_NO_GEMM_8PH=1
run gemm8ph_b A=1
run gemmpp_b CW_NO_GEMM_8PH=1
