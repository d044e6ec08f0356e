#!/bin/bash
# experiments build: skinny planes with slice statistics (mode 2) against the round-3 path, batch 64 and beam 8 x 5
R=$GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=300 -p no:cacheprovider -k "skinny" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x --timeout=300 -p no:cacheprovider -k "large_batch" 2>&1 | tail -3
for cfg in "b64 0 --batch 64 --tokens 24" "b64 2 --batch 64 --tokens 24" "beam 0 --batch 8 --tokens 24 --num-beams 5" "beam 2 --batch 8 --tokens 24 --num-beams 5"; do
  set -- $cfg; TAG=$1_sk$2; export CW_SKINNY=$2; shift; shift
  mkdir -p $R/gpurun_out/prof_$TAG
  cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o run -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-longform --no-config3 "$@" > $R/gpurun_out/prof_bench_$TAG.json 2>/dev/null
  cd $R
  DB=$(ls gpurun_out/prof_$TAG/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python profiles/summarize.py $DB > gpurun_out/kernel_stats_$TAG.txt
  rm -rf gpurun_out/prof_$TAG
  echo "== $TAG"; grep -E "gemv_prep|gemv_mt|skinny|attn_decode|attn_cross" gpurun_out/kernel_stats_$TAG.txt | cut -c1-70,100-160
  python - <<PY
import json
d=json.loads(open("gpurun_out/prof_bench_$TAG.json").read().strip().splitlines()[-1])
print("   step", round(d["ms_per_step"],1), "decode per step", round(d["stage_roofline"]["decode_step"]["ms_per_step"],4))
PY
done
