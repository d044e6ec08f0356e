"""Block start-time distribution of the decode GEMVs (development aid; library built with EXTRA=-DCW_PHASE_TIMING)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crisperwhisper_amd import synthetic as syn
from crisperwhisper_amd.engine import Engine
g, v = syn.large_v3_geometry()
spec = syn.model_spec(g, v, 15)
eng = Engine(spec, dtype="bf16", max_batch=8)
eng.lib.cw_debug_phases.argtypes = [C.c_void_p]
for which, name, nblk in [(0, "fc1", 160), (3, "qkv", 240), (4, "q_c", 80), (5, "fc2", 160), (2, "o-proj", 160)]:
    ms, by = eng.time_kernel(which, 8, 50)
    buf = np.zeros(512 * 8, np.uint64)
    eng.lib.cw_debug_phases(buf.ctypes.data_as(C.c_void_p))
    ph = buf.reshape(512, 8)[:nblk, :6].astype(np.int64)
    t0 = ph[:, 0].min()
    rel = (ph - t0) / 100.0   # 100 MHz -> us
    ent = rel[:, 0]
    print(f"{name}: avg kernel {ms*1e3:.2f} us; entry us: min {ent.min():.2f} p25 {np.percentile(ent,25):.2f} p50 {np.percentile(ent,50):.2f} p75 {np.percentile(ent,75):.2f} max {ent.max():.2f}; end (reduce-done) max {rel[:,5].max():.2f}")
    print("   per-block duration entry->reduce-done: mean %.2f min %.2f max %.2f" % ((rel[:,5]-rel[:,0]).mean(), (rel[:,5]-rel[:,0]).min(), (rel[:,5]-rel[:,0]).max()))
    for x in range(8):
        e = np.sort(ent[x::8])
        print(f"   xcd {x}: entries {np.round(e[:6],2)} ... {np.round(e[-3:],2)}")
    print("   phases mean us (entry, loads-issued, LN-done, post-barrier, mfma-done, reduce-done):", np.round(rel.mean(0), 2))
