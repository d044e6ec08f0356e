#!/bin/bash
# beam-search decode A/B: kernel table of `bench.py --num-beams 5` per variant; usage: run_gpu_beam_ab.sh "NAME ENV=VAL" ...
export PYTHONUNBUFFERED=1
for v in "$@"; do
  set -- $v
  timeout 250 bash tools/ab/run_gpu_prof2.sh $1 $2 -- --num-beams 5 > /dev/null 2>&1
  echo "== $v"; grep -E "cross_|gemv_rows|gemv_mt|gemv_prep|attn_decode|rows_prep" gpurun_out/prof_$1.txt | cut -c9-72,100-135
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_$1.log | head -1
done
