#!/bin/bash
# tests + profile in one box visit; args: batch tokens
B=${1:-8}; T=${2:-32}
mkdir -p gpurun_out/prof
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o run -- python $R/bench.py --batch $B --tokens $T --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
grep -v "^W2026\|^I2026\|^E2026" $R/gpurun_out/prof_bench.log | tail -2 | cut -c1-2500
