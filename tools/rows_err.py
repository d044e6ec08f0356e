"""Teacher-forced logits of the bf16 engine's two 17..64-row paths against the f32 engine (same weights, 2+2-layer large-v3
shapes): which path sits closer to the f32 arithmetic."""
import numpy as np
from crisperwhisper_amd import synthetic as syn
from crisperwhisper_amd.engine import Engine

g, v = syn.large_v3_geometry()
g.enc_layers = g.dec_layers = 2
spec = syn.model_spec(g, v, n_align=15)
spec.alignment_heads = [[l, h] for l in range(2) for h in (0, 3, 7, 19)]
rows, T = 20, 12
clips = [syn.synth_audio(500 + i, 480000 - 5000 * i, ("noise", "chirp", "mixed")[i % 3]) for i in range(rows)]
rng = np.random.default_rng(3)
ids = np.concatenate([[v.sot, v.lang_id("en"), v.transcribe], [v.timestamp_begin], rng.integers(300, 50000, T - 4)])
forced = np.full((rows, T), -1, np.int32); forced[:, 3:] = ids[3:]
prompt = np.tile(ids[None, :3], (rows, 1))
for wname, W in (("random", syn.random_weights(g, seed=11)),
                 ("aligned", {n: syn.weight_tensor(g, n, s, 0, "aligned") for n, s in syn.weight_shapes(g).items()})):
    out = {}
    for dt, modes in (("f32", (1,)), ("bf16", (1, 0))):
        eng = Engine(spec, dtype=dt, max_batch=rows)
        eng.load_state_dict(W)
        eng.mel(clips)
        eng.encode(list(range(rows)), [0] * rows, [3000] * rows)
        for mode in modes:
            if dt == "bf16":
                eng._chk(eng.lib.cw_set_option(eng.ctx, b"rows_ln", mode))
            cap = eng.capture_logits(rows, T)
            eng.decode(prompt, max_length=T, forced=forced)
            out[(dt, mode)] = cap[:T - 3].copy()
            eng.stop_capture()
        eng.close()
    ref = out[("f32", 1)]
    for mode in (1, 0):
        d = out[("bf16", mode)] - ref
        print(wname, "rows_ln", mode, "max rel", np.abs(d).max() / np.abs(ref).max(), "rms rel", np.sqrt((d ** 2).mean()) / np.sqrt((ref ** 2).mean()))
