#!/bin/bash
mkdir -p gpurun_out/prof
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o run -- python $R/bench.py --batch 8 --tokens 128 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
tail -2 $R/gpurun_out/prof_bench.log
ls -R $R/gpurun_out/prof | head -20
