"""Phase stamps of mlp_chain_kernel (csrc/declayer.hip; library built with EXTRA=-DCW_PHASE_TIMING): wave 0 of blocks 0..255
(160 fc1 blocks, the first 96 fc2 blocks).  usage: python tools/mlp_phase_probe.py [rows]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crisperwhisper_amd import synthetic as syn
from crisperwhisper_amd.engine import Engine
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g, v = syn.large_v3_geometry()
spec = syn.model_spec(g, v, 15)
eng = Engine(spec, dtype="bf16", max_batch=rows)
for name, shape in syn.weight_shapes(g).items():
    eng.load_tensor(name, syn.weight_tensor(g, name, shape, 0, "aligned"))
eng.check_weights()
eng.lib.cw_debug_dl_phases.argtypes = [C.c_void_p]
st = eng.time_decode_stages(rows, 64)
for s_ in st:
    print("%-80s %7.2f us" % (s_["kernel"][:80], s_["avg_ms"] * 1e3))
idx = [s_["stage"] for s_ in st if "mlp_chain" in s_["kernel"]]
if not idx:
    raise SystemExit("the layer has no mlp_chain launch")
ms, by, kind, ns = C.c_float(0), C.c_double(0), C.c_int32(0), C.c_int32(0)
eng._chk(eng.lib.cw_time_decode_stage(eng.ctx, rows, idx[0], 64, C.byref(ms), C.byref(by), C.byref(kind), C.byref(ns)))
buf = np.zeros(256 * 3 * 16, np.uint64)
eng.lib.cw_debug_dl_phases(buf.ctypes.data_as(C.c_void_p))
ph = buf.reshape(256, 3, 16)[:, 0, :].astype(np.int64)
t0 = ph[:, 0].min()
def show(title, blk, names):
    print(title)
    for i, n in names.items():
        col = ph[blk, i]
        ok = col > 0
        if ok.sum() == 0:
            continue
        r = (col[ok] - t0) / 100.0
        print(f"   {n:34s} n={ok.sum():3d}  min {r.min():6.2f}  p50 {np.median(r):6.2f}  p90 {np.percentile(r, 90):6.2f}  max {r.max():6.2f}")
show("fc1 blocks (0..159)", slice(0, 160), {0: "entry", 1: "loads issued", 2: "LN done, rows in LDS", 3: "after barrier", 4: "MFMA done", 5: "rows stored", 6: "drained + barrier"})
show("fc2 blocks (160..255)", slice(160, 256), {0: "entry", 1: "weights issued", 7: "wave 0: flags ready", 8: "after barrier", 9: "rows loaded into LDS", 10: "MFMA done", 11: "atomics issued"})
eng.close()
