#!/bin/bash
# kernel-trace of a short bench run with and without an env switch ($1), summaries to gpurun_out/prof_ab_{off,on}.txt
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in off on; do
  rm -rf /tmp/prof_ab
  if [ $mode = on ]; then export $1=1; else unset $1; fi
  timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_ab -o run -- python $R/bench.py --batch ${2:-8} --tokens 4 --steps 1 --warmup 0 --no-cpu-baseline --kernel-iters 3 > /dev/null 2>&1
  python $R/profiles/summarize.py /tmp/prof_ab/run_results.db > $R/gpurun_out/prof_ab_$mode.txt
done
