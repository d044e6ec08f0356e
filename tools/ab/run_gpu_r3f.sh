#!/bin/bash
export PYTHONUNBUFFERED=1
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3f_bench_full.json 2> gpurun_out/r3f_bench_full.err ) 2>&1 | grep real
python - <<'P'
import json
d=json.load(open("gpurun_out/r3f_bench_full.json"))
print(d["value"], d["ms_per_step"], d["rtf"], d["stage_ms_per_step"], d["parity"]["ok"])
print(d["roofline"])
print({k:(round(v.get("frac_of_8TBps", v.get("frac_of_2500TFps", 0)),4)) for k,v in d["stage_roofline"].items() if isinstance(v, dict)})
print(d["longform"]["wall_s"], d["longform"]["words"]); print(d["cpu_baseline"]["value"], d["cpu_baseline"].get("rtf"))
print({m: (round(v["ms_per_step"],1), v["golden_clips_identical_text"]) for m, v in d["config3"]["modes"].items()})
P
bash tools/ab/run_gpu_pmc.sh gpurun_out/r03_pmc_traffic_B8.json > /dev/null 2>&1; head -c 1500 gpurun_out/r03_pmc_traffic_B8.json
