"""Phase stamps (gemm.hip: PH, library built with EXTRA=-DCW_PHASE_TIMING) of the decode GEMV launches AS THE STEP ISSUES THEM:
every gemv2_bf16_kernel stage of the decoder layer is timed through cw_time_decode_stage and the stamps of its last launch are
printed.  usage: python tools/gemv_stage_phase_probe.py [rows]"""
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
eng.lib.cw_debug_phases.argtypes = [C.c_void_p]
st = eng.time_decode_stages(rows, 64)
ms, by, kind, ns = C.c_float(0), C.c_double(0), C.c_int32(0), C.c_int32(0)
for s_ in st:
    if s_["stage"] < 0 or not any(k in s_["kernel"] for k in ("out-projection (combines", "fc1", "fc2")):
        continue
    eng._chk(eng.lib.cw_time_decode_stage(eng.ctx, rows, s_["stage"], 64, C.byref(ms), C.byref(by), C.byref(kind), C.byref(ns)))
    buf = np.zeros(512 * 8, np.uint64)
    eng.lib.cw_debug_phases(buf.ctypes.data_as(C.c_void_p))
    ph = buf.reshape(512, 8)[:, :6].astype(np.int64)
    ph = ph[ph[:, 0] > 0]
    last = ph[:, 0].max()
    ph = ph[ph[:, 0] > last - 2000]          # the blocks of the last launch (20 us window)
    t0 = ph[:, 0].min()
    rel = (ph - t0) / 100.0
    print("%-70s %6.2f us per launch, %d blocks stamped" % (s_["kernel"][:70], ms.value * 1e3, len(ph)))
    print("   phase:  entry  loads-accepted  rows-in-LDS  post-barrier  MFMA-done  reduce-done")
    print("   p50 us:", np.round(np.median(rel, 0), 2))
    print("   max us:", np.round(rel.max(0), 2))
eng.close()
