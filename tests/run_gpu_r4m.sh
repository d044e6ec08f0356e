#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python bench.py --no-cpu-baseline --no-longform > gpurun_out/r4m_bench.json 2> gpurun_out/r4m_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4m_bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "parity", d["parity"]["clips_with_identical_text"], d["parity"]["words_identical_and_within_20ms"], d["parity"]["ok"])
for m,v in d["config3"]["modes"].items(): print("config3", m, round(v["ms_per_step"],1), v["golden_clips_identical_text"], v["golden_words_within_20ms"], v.get("golden_clips_differing"), v.get("parity_ok"))
PY
