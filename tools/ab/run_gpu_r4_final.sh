#!/bin/bash
# round-4 evidence: default bench line, rocprofv3 kernel tables (B = 8 headline, B = 64, beam 8 x 5), PMC traffic
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export PYTHONUNBUFFERED=1
cd $R
timeout 900 python bench.py > gpurun_out/r04_final_bench.json 2> gpurun_out/r04_final_bench.err; echo "bench rc=$?"
prof() {  # TAG, bench args...
  TAG=$1; shift
  mkdir -p $R/gpurun_out/prof_$TAG
  cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o run -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-longform --no-config3 "$@" > $R/gpurun_out/prof_bench_$TAG.json 2>/dev/null
  cd $R
  DB=$(ls gpurun_out/prof_$TAG/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python profiles/summarize.py $DB > gpurun_out/kernel_stats_$TAG.txt
  rm -rf gpurun_out/prof_$TAG
}
prof r04_B8 --batch 8 --tokens 128
prof r04_B64 --batch 64 --tokens 128
prof r04_beam5 --batch 8 --tokens 128 --num-beams 5
bash tools/ab/run_gpu_pmc.sh gpurun_out/r04_pmc_traffic_B8.json > /dev/null 2>&1
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_final_bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "rtf", d["rtf"])
print("stages", d["stage_ms_per_step"])
print("roofline", {k: d["roofline"][k] for k in ("achieved","frac","avg_launch_ms","traffic")})
for k,v in d["stage_roofline"].items(): print(" ", k, {kk: (round(vv,5) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk not in ("note",)})
print("parity", d["parity"]["clips_with_identical_text"], d["parity"]["words_identical_and_within_20ms"], d["parity"]["ok"])
for m,v in d["config3"]["modes"].items(): print("config3", m, round(v["ms_per_step"],1), v["golden_clips_identical_text"], v["golden_words_within_20ms"], round(v["encoder_frac_of_peak"],3), v.get("parity_ok"))
print("longform", d["longform"]["wall_s"], d["longform"]["words"])
cb=d["cpu_baseline"]; print("cpu", cb["value"], cb.get("samples_aligned_words_per_s"), cb["cores"], cb.get("cpu_model"))
PY
for t in r04_B8 r04_B64 r04_beam5; do echo "== $t"; head -14 gpurun_out/kernel_stats_$t.txt | cut -c1-160; done
