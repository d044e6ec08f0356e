"""Which of two 16-bit engine variants is closer to the f32 engine where they differ (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crisperwhisper_amd import synthetic as syn
from crisperwhisper_amd.engine import Engine
rows, layers = 8, 4
g, v = syn.large_v3_geometry()
g.enc_layers, g.dec_layers = 1, layers
spec = syn.model_spec(g, v, n_align=15)
spec.alignment_heads = [[l, h] for l in range(layers) for h in (0, 3, 7, 19)][:15]
W = syn.random_weights(g, seed=21)
T = g.max_target_positions - 4
clips = [syn.synth_audio(700 + i, 480000 - 20000 * i, ("noise", "chirp", "mixed")[i % 3]) for i in range(rows)]
prompt = np.tile(np.array([[v.sot, v.lang_id("en"), v.transcribe]], np.int32), (rows, 1))
res = {}
forced = None
for name, env, dt in (("launches", {"CW_NO_QKV_SELF": "1"}, "bf16"), ("fused", {}, "bf16"), ("f32", {}, "f32")):
    os.environ.update(env)
    try:
        eng = Engine(spec, dtype=dt, max_batch=rows)
    finally:
        for k in env:
            os.environ.pop(k, None)
    eng.load_state_dict(W)
    eng.mel(clips)
    eng.encode(list(range(rows)), [0] * rows, [3000] * rows)
    seqs, lens, _ = eng.decode(prompt, max_length=T, min_new_tokens=T - 3, forced=forced)
    if forced is None:                      # teacher-force the later engines on the first one's tokens
        forced = np.full((rows, T), -1, np.int32); forced[:, 3:] = seqs[:, 3:T]
    res[name] = eng.alignment(rows, T - 1).copy()
    eng.close()
d = res["fused"] != res["launches"]
print("differing:", int(d.sum()))
for b in range(rows):
    for lo, hi in ((0, 128), (128, 313), (313, 443)):
        sl = (b, slice(4, 15), slice(lo, hi))
        ef = np.abs(res["fused"][sl] - res["f32"][sl]).mean()
        el = np.abs(res["launches"][sl] - res["f32"][sl]).mean()
        print(f"row {b} pos {lo}-{hi}: mean |fused - f32| {ef:.3e}   mean |launches - f32| {el:.3e}")
