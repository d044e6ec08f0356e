#!/bin/bash
# PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy: three separate rocprofv3 runs, counters only) + workload sweep.
mkdir -p gpurun_out/pmc2
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
ARGS="--batch 8 --tokens 4 --steps 1 --warmup 0 --no-cpu-baseline --no-longform --kernel-iters 3"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc2 -o fetch -- python $R/bench.py $ARGS > $R/gpurun_out/pmc2_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc2 -o write -- python $R/bench.py $ARGS > $R/gpurun_out/pmc2_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc2 -o mfma -- python $R/bench.py $ARGS > $R/gpurun_out/pmc2_mfma.log 2>&1
cd $R
python profiles/summarize_pmc.py gpurun_out/pmc2/fetch_results.db gpurun_out/pmc2/write_results.db gpurun_out/pmc2/mfma_results.db "python bench.py $ARGS" > gpurun_out/r02_pmc_traffic_B8.json && head -c 1500 gpurun_out/r02_pmc_traffic_B8.json
rm -f gpurun_out/pmc2/*.db
for t in 64 224; do
  python bench.py --tokens $t --steps 3 --warmup 1 --no-cpu-baseline --no-longform 2>/dev/null | tail -1 > gpurun_out/r02_sweep_T$t.json
done
for b in 1 16 32 64; do
  timeout 600 python bench.py --batch $b --steps 2 --warmup 1 --no-cpu-baseline --no-longform 2>gpurun_out/r02_sweep_B$b.err | tail -1 > gpurun_out/r02_sweep_B$b.json
done
timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-longform --cross-kv fp8 2>/dev/null | tail -1 > gpurun_out/r02_sweep_B64_fp8kv.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02_sweep_*.json")):
    try:
        j = json.loads(open(f).read())
        print(f, round(j["ms_per_step"], 1), "ms  rtf", round(j["rtf"], 5), " words/s", round(j["value"], 1), j["stage_ms_per_step"], j.get("parity", {}) and j["parity"].get("clips_with_identical_text"))
    except Exception as e:
        print(f, "FAILED", e)
PY
