#!/bin/bash
# counters-only passes; usage: run_gpu_pmc2.sh NAME "CTR1 CTR2 ..." [ENV=VAL ...]
NAME=$1; CTRS=$2; shift; shift
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_$NAME
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 600 rocprofv3 --pmc $CTRS -d $R/gpurun_out/pmc_$NAME -o run -- python $R/bench.py --batch 8 --tokens 6 --steps 1 --warmup 0 --no-cpu-baseline --no-longform --no-config3 --no-rccl --kernel-iters 3 > $R/gpurun_out/pmc_$NAME.log 2>&1
cd $R
python tools/pmc_table.py gpurun_out/pmc_$NAME/run_results.db > gpurun_out/pmc_$NAME.txt 2>&1
rm -rf gpurun_out/pmc_$NAME
