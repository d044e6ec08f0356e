mkdir -p gpurun_out
B="--steps 2 --warmup 1 --no-cpu-baseline --no-longform --no-config3"
for v in nt1 nt2 ksplit; do
  E="X=1"; [ $v = ksplit ] && E="CW_NO_OWN_COLS=1"; [ $v = nt2 ] && E="CW_OWN_NT=2"
  env $E python bench.py $B --batch 64 > gpurun_out/r06_own2_b64_$v.json 2> gpurun_out/r06_own2_b64_$v.err
  env $E python bench.py $B --num-beams 5 > gpurun_out/r06_own2_beam_$v.json 2> gpurun_out/r06_own2_beam_$v.err
  env $E python bench.py $B --batch 40 > gpurun_out/r06_own2_b40_$v.json 2> gpurun_out/r06_own2_b40_$v.err
done
python - <<'PY'
import json
for n in ("b64_nt1","b64_nt2","b64_ksplit","b40_nt1","b40_nt2","b40_ksplit","beam_nt1","beam_nt2","beam_ksplit"):
    try:
        l=json.load(open(f"gpurun_out/r06_own2_{n}.json"))
        p=l.get("parity") or {}
        print(n, "ms/step", round(l["ms_per_step"],1), "decode step ms", round(l["stage_roofline"]["decode_step"]["ms_per_step"],4),
              "parity", p.get("clips_with_identical_text"), p.get("words_identical_and_within_20ms"), "passes", l["passes_per_step"])
    except Exception as e:
        print(n, "failed", e)
PY
