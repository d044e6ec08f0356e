#!/bin/bash
R=$GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
TAG=mel
mkdir -p $R/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o run -- python $R/bench.py --batch 8 --tokens 8 --steps 2 --warmup 1 --no-cpu-baseline --no-longform --no-config3 > $R/gpurun_out/prof_bench_$TAG.log 2>&1
cd $R
DB=$(ls gpurun_out/prof_$TAG/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python profiles/summarize.py $DB > gpurun_out/kernel_stats_$TAG.txt && grep -E "mel|layernorm|kernel  " gpurun_out/kernel_stats_$TAG.txt | cut -c1-170
rm -rf gpurun_out/prof_$TAG
