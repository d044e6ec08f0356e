#!/bin/bash
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_kernels.py -x -q -m gpu -k "fused_decoder_stage or large_batch_decode or reproducible or gemv or bench_shape or large_geometry or teacher_forced or beam_search_bf16 or bench_model_beam or pipeline_bf16 or from_checkpoint" 2>&1 | tail -8
bash tests/run_gpu_prof2.sh r3e A=1 -- > /dev/null 2>&1
grep -E "cross_|gemv_stack|gemv2_bf16|attn_decode" gpurun_out/prof_r3e.txt | cut -c1-75,100-160
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --no-rccl --kernel-iters 50"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $ARGS > gpurun_out/r3e_$name.json 2> gpurun_out/r3e_$name.err;
  python - <<P
import json
try:
    d=json.load(open("gpurun_out/r3e_$name.json")); print("$name", round(d["ms_per_step"],1), d["stage_ms_per_step"], round(d["stage_roofline"]["decode_step"]["ms_per_step"],4), d["parity"]["ok"])
except Exception as e: print("$name failed", e); print(open("gpurun_out/r3e_$name.err").read()[-800:])
P
}
run pack A=1
run nopack CW_NO_WPACK=1
