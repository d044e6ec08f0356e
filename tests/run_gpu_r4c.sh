#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=300 -p no:cacheprovider -k "skinny" 2>&1 | tail -25 > gpurun_out/r4c_kernel.log
tail -5 gpurun_out/r4c_kernel.log
timeout 300 python tools/skinny_ablate.py 64 > gpurun_out/r4c_ablate.txt 2>&1
cat gpurun_out/r4c_ablate.txt
