#!/bin/bash
# rocprofv3 kernel trace + stats of one bench step with extra bench arguments: usage run_gpu_prof_args.sh TAG [bench args...]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o run -- python $R/bench.py --batch 8 --tokens 128 --steps 1 --warmup 1 --no-cpu-baseline --no-longform "$@" > $R/gpurun_out/prof_bench_$TAG.log 2>&1
cd $R
DB=$(ls gpurun_out/prof_$TAG/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python profiles/summarize.py $DB > gpurun_out/kernel_stats_$TAG.txt && head -26 gpurun_out/kernel_stats_$TAG.txt | cut -c1-175
rm -rf gpurun_out/prof_$TAG
