#!/bin/bash
# round 4, call A: skinny-M decoder projections -- kernel test, K split sweep, then the tests that exercise 17..64 decoder rows
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=300 -p no:cacheprovider -k "skinny" 2>&1 | tail -25 > gpurun_out/r4a_kernel.log
tail -12 gpurun_out/r4a_kernel.log
timeout 400 python tools/skinny_bench.py 40 64 > gpurun_out/r4a_sweep.txt 2>&1
cat gpurun_out/r4a_sweep.txt
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -k "large_batch or beam" 2>&1 | tail -30 > gpurun_out/r4a_rows.log
tail -15 gpurun_out/r4a_rows.log
