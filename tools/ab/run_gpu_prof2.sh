#!/bin/bash
# kernel-trace profile of a short bench run; usage: run_gpu_prof2.sh NAME [ENV=VAL ...] -- extra bench args
NAME=$1; shift
ENVS=(); while [ "$1" != "--" ] && [ -n "$1" ]; do ENVS+=("$1"); shift; done; shift
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_$NAME
cd /tmp && export TMPDIR=/tmp
env "${ENVS[@]}" timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_$NAME -o run -- python $R/bench.py --batch 8 --tokens 128 --steps 1 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --no-rccl --kernel-iters 20 "$@" > $R/gpurun_out/prof_$NAME.log 2>&1
cd $R
python profiles/summarize.py gpurun_out/prof_$NAME/run_results.db > gpurun_out/prof_$NAME.txt
rm -rf gpurun_out/prof_$NAME
head -24 gpurun_out/prof_$NAME.txt | cut -c1-175
