"""Block shapes of the 33..64-row decode GEMVs (gemm.hip: gemv_mt_kernel<.., NT> + row groups): every launch of the decoder layer
timed through the step's own launch code (cw_time_decode_stage) for variant 0 (16 columns x all row tiles), 1 (two row groups),
2 (two row groups x two column tiles).  usage: python tools/mt_variant_probe.py [rows]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crisperwhisper_amd import synthetic as syn
from crisperwhisper_amd.engine import Engine
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g, v = syn.large_v3_geometry()
spec = syn.model_spec(g, v, 15)
eng = Engine(spec, dtype="bf16", max_batch=rows)
for name, shape in syn.weight_shapes(g).items():
    eng.load_tensor(name, syn.weight_tensor(g, name, shape, 0, "aligned"))
eng.check_weights()
res = {}
for var in (0, 1, 2, 0):
    assert eng.lib.cw_test_set_option(b"mt_variant", var) == 0
    res.setdefault(var, []).append(eng.time_decode_stages(rows, 64))
eng.lib.cw_test_set_option(b"mt_variant", -1)
base = res[0][0]
print("%-72s %9s %9s %9s %9s" % (f"launch ({rows} rows; a 'launch' of a LayerNorm projection = preparation + GEMV)", "v0", "v1", "v2", "v0 again"))
for i, s_ in enumerate(base):
    print("%-72s %9.2f %9.2f %9.2f %9.2f" % (s_["kernel"][:72], s_["avg_ms"] * 1e3, res[1][0][i]["avg_ms"] * 1e3, res[2][0][i]["avg_ms"] * 1e3, res[0][1][i]["avg_ms"] * 1e3))
eng.close()
