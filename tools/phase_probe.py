"""Phase timeline of the decode GEMV blocks (development aid).  Needs a library built with
  make -C crisperwhisper_amd/csrc clean all EXTRA=-DCW_PHASE_TIMING   (see gemm.hip: PH / cw_debug_phases)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crisperwhisper_amd import synthetic as syn
from crisperwhisper_amd.engine import Engine
g, v = syn.large_v3_geometry()
spec = syn.model_spec(g, v, 15)
eng = Engine(spec, dtype="bf16", max_batch=8)
for which, name, nblk in [(0, "fc1", 320), (3, "qkv", 240), (4, "q_c", 80), (5, "fc2", 320), (2, "o-proj", 160)]:
    ms, by = eng.time_kernel(which, 8, 50)
    buf = np.zeros(512 * 8, np.uint64)
    eng.lib.cw_debug_phases.argtypes = [C.c_void_p]
    eng.lib.cw_debug_phases(buf.ctypes.data_as(C.c_void_p))
    ph = buf.reshape(512, 8)[:nblk, :6].astype(np.int64)
    t0 = ph[:, 0].min()
    rel = (ph - t0) * 10.0 / 1000.0   # 100 MHz -> us
    print(name, "avg kernel %.2f us" % (ms * 1e3))
    print("  phase: entry, loads-issued, pre-barrier(LN done), post-barrier, mfma-done(weights arrived), reduce-done")
    print("  mean us:", np.round(rel.mean(0), 2))
    print("  min  us:", np.round(rel.min(0), 2))
    print("  max  us:", np.round(rel.max(0), 2))
