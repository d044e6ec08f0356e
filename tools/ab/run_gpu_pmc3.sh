#!/bin/bash
# counters-only pass of a short bench run; usage: run_gpu_pmc3.sh NAME "CTR" [ENV=VAL ...] -- extra bench args
# ONE counter per pass: FETCH_SIZE together with WRITE_SIZE exceeds what the hardware collects at once -- rocprofv3 aborts while
# the engine is being created and the process then sits until the timeout (which is why that is short here).
NAME=$1; CTRS=$2; shift; shift
ENVS=(); while [ "$1" != "--" ] && [ -n "$1" ]; do ENVS+=("$1"); shift; done; shift
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_$NAME
cd /tmp && export TMPDIR=/tmp
env "${ENVS[@]}" timeout 120 rocprofv3 --pmc $CTRS -d $R/gpurun_out/pmc_$NAME -o run -- python $R/bench.py --batch 8 --tokens 6 --steps 1 --warmup 0 --no-cpu-baseline --no-longform --no-config3 --no-rccl --kernel-iters 3 "$@" > $R/gpurun_out/pmc_$NAME.log 2>&1
cd $R
python tools/pmc_table.py gpurun_out/pmc_$NAME/run_results.db > gpurun_out/pmc_$NAME.txt 2>&1
rm -rf gpurun_out/pmc_$NAME
