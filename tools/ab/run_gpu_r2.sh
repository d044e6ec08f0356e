#!/bin/bash
# One box visit: GPU test suite, default bench line (with the reference cpu_baseline + longform leg), A/B bench lines for
# the env toggles given as arguments ("CW_NO_FUSE_SELF=1" ...), and a rocprofv3 kernel trace of the bench step.
# usage: tools/ab/run_gpu_r2.sh <tag> [ENV=VAL ...]
TAG=${1:-r02}; shift
mkdir -p gpurun_out/prof_$TAG
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 -p no:cacheprovider --durations=8 ${PYTEST_ARGS} 2>&1 | tail -40 > gpurun_out/pytest_$TAG.log
  tail -14 gpurun_out/pytest_$TAG.log
fi
if [ -z "$SKIP_BENCH" ]; then
  timeout 900 python bench.py --steps ${STEPS:-5} --warmup 2 ${BENCH_ARGS} > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
  tail -c 3000 gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
fi
for kv in "$@"; do
  name=$(echo $kv | tr '= ' '__')
  env $kv timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-longform > gpurun_out/bench_${TAG}_$name.json 2> gpurun_out/bench_${TAG}_$name.err
  python - <<P
import json
try:
    d = json.loads(open("gpurun_out/bench_${TAG}_$name.json").read().strip().splitlines()[-1])
    print("$kv", "ms_per_step", round(d["ms_per_step"], 1), d["stage_ms_per_step"])
except Exception as e:
    print("$kv failed", e)
P
done
if [ -z "$SKIP_PROF" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o run -- python $R/bench.py --batch 8 --tokens 128 --steps 1 --warmup 1 --no-cpu-baseline --no-longform > $R/gpurun_out/prof_bench_$TAG.log 2>&1
  cd $R
  DB=$(ls gpurun_out/prof_$TAG/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python profiles/summarize.py $DB > gpurun_out/kernel_stats_$TAG.txt && head -24 gpurun_out/kernel_stats_$TAG.txt | cut -c1-175
fi
