"""Launch-boundary floor: graph vs eager, development aid."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crisperwhisper_amd import synthetic as syn
from crisperwhisper_amd.engine import Engine
g, v = syn.tiny_geometry()
eng = Engine(syn.model_spec(g, v, 3), dtype="bf16", max_batch=8)
for which, name in ((100, "graph of 260 empty kernels"), (101, "eager 260 empty kernels")):
    ms, _ = eng.time_kernel(which, 8, 260)
    print(f"{name}: {ms*1e3:.2f} us per kernel -> {ms*260:.3f} ms per 260")
