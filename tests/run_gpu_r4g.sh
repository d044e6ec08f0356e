#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r4g_pytest.log
tail -12 gpurun_out/r4g_pytest.log
timeout 900 python bench.py > gpurun_out/r4g_bench.json 2> gpurun_out/r4g_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4g_bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "stages", d["stage_ms_per_step"])
print("parity", d["parity"])
print("decode", d["stage_roofline"]["decode_step"]["ms_per_step"], "enc frac", d["stage_roofline"]["encoder"]["frac_of_2500TFps"])
c3=d.get("config3",{})
for m,v in (c3.get("modes") or {}).items(): print("config3", m, v["ms_per_step"], v["golden_clips_identical_text"], v["golden_words_within_20ms"], v.get("golden_clips_differing"))
print("longform", d["longform"]["wall_s"] if d.get("longform") else None)
PY
