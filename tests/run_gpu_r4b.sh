#!/bin/bash
# round 4, call B: ablation of the skinny launches (debug build) + accuracy of the 17..64-row paths against the f32 engine
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python tools/skinny_ablate.py 64 > gpurun_out/r4b_ablate.txt 2>&1
cat gpurun_out/r4b_ablate.txt
timeout 600 python tools/rows_err.py > gpurun_out/r4b_rows_err.txt 2>&1
tail -12 gpurun_out/r4b_rows_err.txt
