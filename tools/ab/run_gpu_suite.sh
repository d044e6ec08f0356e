#!/bin/bash
# Runs the GPU suite on a gpurun box and leaves logs under gpurun_out/.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/gpu.txt
nproc >> gpurun_out/gpu.txt; lscpu | grep "Model name" >> gpurun_out/gpu.txt
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -5 gpurun_out/smoke.log
timeout 600 python bench.py --geometry tiny --batch 4 --tokens 32 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_tiny.log 2>&1; tail -3 gpurun_out/bench_tiny.log
timeout 900 python bench.py --batch 2 --tokens 16 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_large_small.log 2>&1; tail -3 gpurun_out/bench_large_small.log
