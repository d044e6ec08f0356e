#!/bin/bash
# Fused out-projection / cross-query stage under beam search (attn_cross_mfma_kernel<.., FUSED>): beam goldens on the new path, then
# bench.py --num-beams 5 (8 items x 5 hypotheses = 40 decoder rows) with and without it on the same box
mkdir -p gpurun_out
python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "beam" 2>&1 | tail -6
B="--steps 3 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --num-beams 5"
python bench.py $B > gpurun_out/r06_beam_fused_a.json 2> gpurun_out/r06_beam_fused_a.err
CW_NO_FUSE_BEAM=1 python bench.py $B > gpurun_out/r06_beam_twelve_a.json 2> gpurun_out/r06_beam_twelve_a.err
python bench.py $B > gpurun_out/r06_beam_fused_b.json 2> gpurun_out/r06_beam_fused_b.err
CW_NO_FUSE_BEAM=1 python bench.py $B > gpurun_out/r06_beam_twelve_b.json 2> gpurun_out/r06_beam_twelve_b.err
python - <<'PY'
import json
for n in ("fused_a","twelve_a","fused_b","twelve_b"):
    try:
        l=json.load(open(f"gpurun_out/r06_beam_{n}.json"))
        print(n, "ms/step", round(l["ms_per_step"],1), "ms per beam step", round(l["stage_roofline"]["decode_step"]["ms_per_step"],4), "words/s", round(l["value"],1))
    except Exception as e:
        print(n, "failed", e)
PY
