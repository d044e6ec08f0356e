#!/bin/bash
# 2 ranks on the single GPU of a gpurun box (gloo for the collective): exercises bench.py's world>1 branch.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 CW_DIST_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1 --warmup 1 --tokens 16 > gpurun_out/bench_dist2.log 2>&1
tail -2 gpurun_out/bench_dist2.log | cut -c1-900
unset CW_DIST_BACKEND
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 1 --warmup 1 --tokens 16 --no-cpu-baseline > gpurun_out/bench_dist1.log 2>&1
tail -1 gpurun_out/bench_dist1.log | cut -c1-300
