#!/bin/bash
# A/B of the key-split count of the decode cross-attention: ATT_NS = 6 (shipped: 960 blocks at 8 rows = 3.75 per CU) against a
# -DATT_NS=8 build (1280 blocks = 5 per CU exactly), same box, headline bench without the side legs
mkdir -p gpurun_out
B="--steps 5 --warmup 2 --no-cpu-baseline --no-longform --no-config3"
python bench.py $B > gpurun_out/r06_ns6_a.json 2> gpurun_out/r06_ns6_a.err
CW_LIB_PATH=$PWD/crisperwhisper_amd/libcw_ns8.so python bench.py $B > gpurun_out/r06_ns8_a.json 2> gpurun_out/r06_ns8_a.err
python bench.py $B > gpurun_out/r06_ns6_b.json 2> gpurun_out/r06_ns6_b.err
CW_LIB_PATH=$PWD/crisperwhisper_amd/libcw_ns8.so python bench.py $B > gpurun_out/r06_ns8_b.json 2> gpurun_out/r06_ns8_b.err
python - <<'PY'
import json
for n in ("ns6_a","ns8_a","ns6_b","ns8_b"):
    try:
        l=json.load(open(f"gpurun_out/r06_{n}.json"))
        print(n, round(l["ms_per_step"],1), round(l["stage_roofline"]["decode_step"]["ms_per_step"],4), l["parity"]["clips_with_identical_text"], l["parity"]["words_identical_and_within_20ms"],
              [(r["kernel"][:28], round(r["avg_launch_ms"]*1e3,2)) for r in [l["roofline"]]+l["roofline_other"]])
    except Exception as e:
        print(n, "failed", e)
PY
