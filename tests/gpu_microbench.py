"""Decode-step kernel microbenchmarks (HIP events): every kernel cycling through the 32 layers' operands (cold, as in the step)
and on one layer's operands (hot: L2 / Infinity-Cache resident) -- development aid."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crisperwhisper_amd import synthetic as syn
from crisperwhisper_amd.engine import Engine
g, v = syn.large_v3_geometry()
spec = syn.model_spec(g, v, 15)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = Engine(spec, dtype="bf16", max_batch=B)
for name, shape in syn.weight_shapes(g).items():
    eng.load_tensor(name, syn.weight_tensor(g, name, shape, 0, "aligned"))
eng.check_weights()
names = ["fc1(LN,GELU)", "cross-attn", "o-proj(resid)", "qkv(LN,cache)", "q_c(LN)", "fc2(resid)", "self-attn@64", "logits(LN)", "empty"]
for w, n in enumerate(names):
    res = []
    for mode in (0, 200):
        best = 1e9
        for rep in range(3):
            ms, by = eng.time_kernel(w + mode, B, 300)
            best = min(best, ms)
        res.append(best * 1e3)
    print(f"{n:16s} cold {res[0]:7.2f} us   hot {res[1]:7.2f} us")
