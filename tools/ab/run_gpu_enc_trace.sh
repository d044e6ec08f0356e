#!/bin/bash
# Per-dispatch durations of one encoder layer at the bench shape (kernel trace only): usage run_gpu_enc_trace.sh TAG
TAG=${1:-enc}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/enc_trace
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/enc_trace -o t --output-format csv -- python $R/bench.py --batch 8 --tokens 2 --steps 1 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --no-rccl > $R/gpurun_out/enc_trace/log.txt 2>&1
cd $R
python - <<'P' | tee gpurun_out/enc_trace_$TAG.txt
import csv, glob, collections
f = glob.glob("gpurun_out/enc_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last encoder pass: find the last run of 32 attn_encoder kernels
idx = [i for i, n in enumerate(names) if "attn_encoder" in n]
last = idx[-32:]
# layer 10 of that pass: kernels between attention 9 and attention 11
a, b = last[9], last[11]
for r in rows[a + 1:b + 1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{d:9.2f} us  grid {r.get('Grid_Size','?'):>8s}  {r['Kernel_Name'][:110]}")
tot = collections.defaultdict(float); cnt = collections.Counter()
lo, hi = last[0] - 3, last[-1] + 4
for r in rows[lo:hi]:
    k = r["Kernel_Name"][:70]; tot[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; cnt[k] += 1
print("--- whole pass (32 layers)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{v/1e3:8.3f} ms  n={cnt[k]:3d}  avg {v/cnt[k]:8.2f} us  {k}")
span = (int(rows[hi-1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e6
print(f"span {span:.3f} ms, kernel sum {sum(tot.values())/1e3:.3f} ms")
P
rm -rf gpurun_out/enc_trace
