#!/bin/bash
R=$GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -k "large_batch or beam or cross_attention" 2>&1 | tail -30 > gpurun_out/r4f_rows.log
tail -8 gpurun_out/r4f_rows.log
timeout 600 python tools/rows_err.py > gpurun_out/r4f_rows_err.txt 2>&1; tail -16 gpurun_out/r4f_rows_err.txt
for sk in 0 1; do
  export CW_SKINNY=$sk
  TAG=b64_mode$sk
  mkdir -p $R/gpurun_out/prof_$TAG
  cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o run -- python $R/bench.py --batch 64 --tokens 24 --steps 1 --warmup 1 --no-cpu-baseline --no-longform --no-config3 > $R/gpurun_out/prof_bench_$TAG.log 2>&1
  cd $R
  DB=$(ls gpurun_out/prof_$TAG/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python profiles/summarize.py $DB > gpurun_out/kernel_stats_$TAG.txt && head -20 gpurun_out/kernel_stats_$TAG.txt | cut -c1-170
  rm -rf gpurun_out/prof_$TAG
  python - <<PY
import json
d=json.loads(open("gpurun_out/prof_bench_$TAG.log").read().strip().splitlines()[-1])
print("mode $sk B=64 ms_per_step", d["ms_per_step"], d["stage_ms_per_step"], d["parity"])
PY
done
