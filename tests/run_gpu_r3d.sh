#!/bin/bash
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_kernels.py -x -q -m gpu -k "fused_decoder_stage or large_batch_decode or reproducible or gemv or sample or bench_shape or large_geometry" 2>&1 | tail -8
bash tests/run_gpu_prof2.sh r3d A=1 -- > /dev/null 2>&1
grep -E "cross_|gemv_stack|gemv2_bf16|attn_decode|sample" gpurun_out/prof_r3d.txt | cut -c1-75,100-160
for B in 8 16; do
timeout 300 python bench.py --batch $B --steps 3 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --no-rccl --kernel-iters 50 > gpurun_out/r3d_B$B.json 2> gpurun_out/r3d_B$B.err
python - <<P
import json
d=json.load(open("gpurun_out/r3d_B$B.json")); print("B=$B", round(d["ms_per_step"],1), d["stage_ms_per_step"], round(d["stage_roofline"]["decode_step"]["ms_per_step"],4), d["parity"]["ok"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
P
done
CW_NO_FUSE6=1 timeout 300 python bench.py --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --no-rccl --kernel-iters 50 > gpurun_out/r3d_B16_nofuse.json 2>/dev/null
python - <<P
import json
d=json.load(open("gpurun_out/r3d_B16_nofuse.json")); print("B=16 nofuse", round(d["ms_per_step"],1), round(d["stage_roofline"]["decode_step"]["ms_per_step"],4), d["parity"]["ok"])
P
