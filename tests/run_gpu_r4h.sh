#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 -p no:cacheprovider -k "mel" 2>&1 | tail -15 > gpurun_out/r4h_mel.log; tail -5 gpurun_out/r4h_mel.log
for nc in 0 1; do
  if [ $nc = 1 ]; then export CW_NO_STACK_CENTER=1; else unset CW_NO_STACK_CENTER; fi
  timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --timeout=600 -p no:cacheprovider -k "second_weight_seed or fused_decoder_stage" 2>&1 | grep -E "AssertionError|passed|failed|^E  " | head -20 > gpurun_out/r4h_center_$nc.log
  echo "== no_center=$nc"; cat gpurun_out/r4h_center_$nc.log
done
unset CW_NO_STACK_CENTER
for v in 0 1; do
  if [ $v = 1 ]; then export CW_MEL_VALU=1; else unset CW_MEL_VALU; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config3 --no-longform > gpurun_out/r4h_bench_melvalu$v.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4h_bench_melvalu$v.json").read().strip().splitlines()[-1])
print("mel_valu=$v", d["ms_per_step"], d["stage_ms_per_step"], d["stage_roofline"]["mel"], d["parity"]["ok"])
PY
done
