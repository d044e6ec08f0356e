#!/bin/bash
# rocprofv3 kernel tables of the B = 64 step, round-3 path and skinny path
R=$GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
for sk in 0 1; do
  if [ $sk = 0 ]; then export CW_NO_SKINNY=1; else unset CW_NO_SKINNY; fi
  TAG=b64_sk$sk
  mkdir -p $R/gpurun_out/prof_$TAG
  cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o run -- python $R/bench.py --batch 64 --tokens 24 --steps 1 --warmup 1 --no-cpu-baseline --no-longform --no-config3 > $R/gpurun_out/prof_bench_$TAG.log 2>&1
  cd $R
  DB=$(ls gpurun_out/prof_$TAG/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python profiles/summarize.py $DB > gpurun_out/kernel_stats_$TAG.txt && head -22 gpurun_out/kernel_stats_$TAG.txt | cut -c1-170
  rm -rf gpurun_out/prof_$TAG
done
