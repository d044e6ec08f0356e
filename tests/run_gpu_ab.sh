#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for mode in graph nograph; do
  if [ $mode = nograph ]; then export CW_NO_GRAPH=1; else unset CW_NO_GRAPH; fi
  timeout 600 python bench.py --tokens 32 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$mode.log 2>&1
  echo $mode; tail -1 gpurun_out/bench_$mode.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms_per_step'])"
done
