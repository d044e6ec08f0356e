#!/bin/bash
# Column-owning cross-attention out-projection at 33..64 rows (gemv_mt_kernel OWN / LNA): every golden that runs 33..64 decoder rows
# on the new path, then batch 64 (bf16 and e4m3 cache) and beam search 8 x 5 with and without it on the same box
mkdir -p gpurun_out
python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "batch64 or odd_batch or beam or fused_decoder_stage or reproducible or large_batch_decode or row_groups" 2>&1 | tail -8
B="--steps 2 --warmup 1 --no-cpu-baseline --no-longform --no-config3"
for v in own kslit; do
  E=""; [ $v = kslit ] && E="CW_NO_OWN_COLS=1"
  env $E python bench.py $B --batch 64 > gpurun_out/r06_own_b64_$v.json 2> gpurun_out/r06_own_b64_$v.err
  env $E python bench.py $B --batch 64 --cross-kv fp8 > gpurun_out/r06_own_b64fp8_$v.json 2> gpurun_out/r06_own_b64fp8_$v.err
  env $E python bench.py $B --num-beams 5 > gpurun_out/r06_own_beam_$v.json 2> gpurun_out/r06_own_beam_$v.err
done
python - <<'PY'
import json
for n in ("b64_own","b64_kslit","b64fp8_own","b64fp8_kslit","beam_own","beam_kslit"):
    try:
        l=json.load(open(f"gpurun_out/r06_own_{n}.json"))
        p=l.get("parity") or {}
        print(n, "ms/step", round(l["ms_per_step"],1), "decode step ms", round(l["stage_roofline"]["decode_step"]["ms_per_step"],4), "launches", l["stage_roofline"]["decode_step"].get("launches_per_layer"),
              "parity", p.get("clips_with_identical_text"), p.get("words_identical_and_within_20ms"), "words/s", round(l["value"],1))
    except Exception as e:
        print(n, "failed", e)
PY
