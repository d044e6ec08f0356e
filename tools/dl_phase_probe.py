"""Phase stamps of the persistent decoder-layer kernel (csrc/declayer.hip; library built with EXTRA=-DCW_PHASE_TIMING).
usage: python tools/dl_phase_probe.py [rows]      prints per-phase times (us from the first workgroup's entry) over the workgroups"""
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
ms, by = eng.time_kernel(9, rows, 64)
ms8, _ = eng.time_kernel(8, rows, 64)
print(f"epoch bump + persistent stage A: {ms * 1e3:.2f} us per pair; near-empty launch alone {ms8 * 1e3:.2f} us")
buf = np.zeros(256 * 3 * 16, np.uint64)
eng.lib.cw_debug_dl_phases(buf.ctypes.data_as(C.c_void_p))
ph = buf.reshape(256, 3, 16).astype(np.int64)
t0 = ph[:, :, 0].min()
names = {0: "entry", 1: "after first barrier", 2: "requests issued", 3: "tile landed (vmcnt 0)", 4: "chain sync 1", 5: "rows in LDS (sync 2)",
         6: "MFMA done (sync 3)", 7: "published", 8: "first poll back", 9: "query ready", 10: "after query barrier", 11: "item 0 done", 12: "item 1 done"}
for w, wn in ((0, "chain wave 0"), (1, "K/V wave 4 (group 0)"), (2, "K/V wave 8 (group 1)")):
    print(wn)
    for i in sorted(names):
        col = ph[:, w, i]
        ok = col > 0
        if w > 0 and i in (3, 4, 5, 6, 7, 8, 9):
            continue
        if ok.sum() == 0:
            continue
        r = (col[ok] - t0) / 100.0
        print(f"   {names[i]:28s} n={ok.sum():3d}  min {r.min():6.2f}  p50 {np.median(r):6.2f}  p90 {np.percentile(r, 90):6.2f}  max {r.max():6.2f}")
np_ = ph[:, 0, 13]
print("polls per chain wave 0: ", np.bincount(np_[np_ < 64].astype(int)))
eng.close()
