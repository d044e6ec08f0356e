cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1 PYTHONPATH=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() {
  name=$1; shift
  env "$@" timeout 400 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --no-rccl --kernel-iters 20 $EXTRA > gpurun_out/nt_$name.json 2> gpurun_out/nt_$name.err
  python - "$name" <<'P'
import json, sys
try:
    d = json.loads([l for l in open(f"gpurun_out/nt_{sys.argv[1]}.json") if l.startswith("{")][-1])
    print(sys.argv[1], "ms_per_step", round(d["ms_per_step"], 1), "decode ms per token step", round(d["stage_roofline"]["decode_step"]["ms_per_step"], 3), "parity", d.get("parity", {}).get("clips_with_identical_text"), d.get("parity", {}).get("words_identical_and_within_20ms"))
    for r in d.get("roofline_other", []) + [d["roofline"]]:
        print("    ", r["kernel"][:70], round(r["avg_launch_ms"] * 1e3, 2), "us")
except Exception as ex:
    print(sys.argv[1], "ERR", ex)
P
}
EXTRA=""
run nt2 CW_STACK_NT3=2
run nt3 CW_STACK_NT3=3
run nt1 A=1
