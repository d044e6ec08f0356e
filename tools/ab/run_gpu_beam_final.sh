#!/bin/bash
# beam-search decode, same box: round-2 kernels (env switches) against the round-3 ones, unprofiled wall time per beam step
export PYTHONUNBUFFERED=1
run() {
  name=$1; shift
  env "$@" timeout 200 python bench.py --num-beams 5 --steps 2 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --no-rccl --kernel-iters 20 > gpurun_out/beamfin_$name.json 2> gpurun_out/beamfin_$name.err
  python - "$name" <<'P'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/beamfin_{sys.argv[1]}.json") if l.startswith("{")][-1])
print(sys.argv[1], "ms_per_step", round(d["ms_per_step"], 1), "passes", d.get("passes_per_step"), "words/s", round(d["value"], 1),
      "decode ms per beam step", round(d["stage_roofline"]["decode_step"]["ms_per_step"], 3))
P
}
run r2 CW_CROSS_VALU=1 CW_ANC_ATTN_V1=1 CW_MT_NO_PREA=1
run r3 A=1
run r3_noprea CW_MT_NO_PREA=1
run r3_again A=1
