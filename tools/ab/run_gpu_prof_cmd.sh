#!/bin/bash
# rocprofv3 kernel trace of an arbitrary command: usage run_gpu_prof_cmd.sh TAG cmd...   (summary -> gpurun_out/kernel_stats_TAG.txt)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o run -- "$@" > $R/gpurun_out/prof_cmd_$TAG.log 2>&1 )
cd $R
DB=$(ls gpurun_out/prof_$TAG/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python profiles/summarize.py $DB > gpurun_out/kernel_stats_$TAG.txt && head -${HEADN:-22} gpurun_out/kernel_stats_$TAG.txt | cut -c1-175
rm -rf gpurun_out/prof_$TAG
