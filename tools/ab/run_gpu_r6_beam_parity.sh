B="--steps 2 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --num-beams 5"
for v in twelve fused own; do
  E="X=1"; [ $v = twelve ] && E="CW_NO_FUSE_BEAM=1"; [ $v = fused ] && E="CW_NO_OWN_COLS=1"
  for dt in bf16 f16; do
    env $E python bench.py $B --dtype $dt > gpurun_out/r06_beampar_${v}_$dt.json 2> gpurun_out/r06_beampar_${v}_$dt.err
  done
done
python - <<'PY'
import json
for v in ("twelve","fused","own"):
  for dt in ("bf16","f16"):
    try:
        l=json.load(open(f"gpurun_out/r06_beampar_{v}_{dt}.json")); p=l["parity"]
        print(v, dt, "ms/step", round(l["ms_per_step"],1), "passes", l["passes_per_step"], "clips", p["clips_with_identical_text"], "words", p["words_identical_and_within_20ms"], "F1", round(p["timestamp_f1_collar_0.2s"],4))
    except Exception as e: print(v, dt, "failed", e)
PY
