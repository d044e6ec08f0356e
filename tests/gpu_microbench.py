"""Decode-step kernel microbenchmarks (HIP events, hot caches) -- development aid."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crisperwhisper_amd import synthetic as syn
from crisperwhisper_amd.engine import Engine
g, v = syn.large_v3_geometry()
spec = syn.model_spec(g, v, 15)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = Engine(spec, dtype="bf16", max_batch=B)
names = ["fc1(LN,GELU)", "cross-attn", "o-proj(resid)", "qkv(LN,cache)", "q_c(LN)", "fc2(resid)", "self-attn@64", "logits(LN)", "empty"]
res = {}
for rep in range(2):
    for w, n in enumerate(names):
        ms, by = eng.time_kernel(w, B, 300)
        res[n] = (ms * 1e3, by / (ms * 1e-3) / 1e9 if ms > 0 else 0)
for n in names:
    print(f"{n:16s} {res[n][0]:8.2f} us  {res[n][1]:8.0f} GB/s")
