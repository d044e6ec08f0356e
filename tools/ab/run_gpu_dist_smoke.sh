#!/bin/bash
# bench.py's world > 1 branch on the single GPU of a gpurun box: 2 ranks share the GPU, gloo carries the collectives
# (two NCCL ranks cannot share one device).  Covers the weak-scaling main loop AND the configs[2] longform leg
# (chunk-sharded 600 s recording, all-gather of chunk records, seam merge on every rank) -- the code the driver's
# 2 / 4 / 8-GPU runs execute with RCCL.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 CW_DIST_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --tokens 32 > gpurun_out/bench_dist2.log 2> gpurun_out/bench_dist2.err
tail -1 gpurun_out/bench_dist2.log | cut -c1-1500; tail -3 gpurun_out/bench_dist2.err
unset CW_DIST_BACKEND
# launched through torch.distributed.run with one rank: RCCL communicator from the launcher's rendezvous env
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 1 --warmup 1 --tokens 32 --no-cpu-baseline > gpurun_out/bench_dist1.log 2> gpurun_out/bench_dist1.err
tail -1 gpurun_out/bench_dist1.log | cut -c1-600; tail -2 gpurun_out/bench_dist1.err
