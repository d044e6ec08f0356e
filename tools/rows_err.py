"""Teacher-forced logits and alignment rows of the 16-bit engines' 17..64-row paths against the f32 engine (same weights,
2+2-layer large-v3 shapes): which path sits closer to the f32 arithmetic.  Paths: xfull = round 4 default (cross-attention query planes finished inside the full-key cross-attention), skinny = every
LayerNorm projection through csrc/skinny.hip planes + finish,
prep = preparation launch + 16-column GEMV (round 3), rows = the rejected no-preparation variant of round 3."""
import numpy as np
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    for dt, modes in (("f32", ("f32",)), ("bf16", ("xfull", "skinny", "prep", "rows")), ("f16", ("xfull", "skinny", "prep"))):
        eng = Engine(spec, dtype=dt, max_batch=rows)
        eng.load_state_dict(W)
        eng.mel(clips)
        eng.encode(list(range(rows)), [0] * rows, [3000] * rows)
        for mode in modes:
            if dt != "f32":
                eng._chk(eng.lib.cw_set_option(eng.ctx, b"skinny", {"xfull": 1, "skinny": 2}.get(mode, 0)))
                eng._chk(eng.lib.cw_set_option(eng.ctx, b"rows_ln", int(mode == "rows")))
            cap = eng.capture_logits(rows, T)
            eng.decode(prompt, max_length=T, forced=forced)
            out[(dt, mode)] = (cap[:T - 3].copy(), eng.alignment(rows, T - 1).copy())
            eng.stop_capture()
        eng.close()
    ref, aref = out[("f32", "f32")]
    for key in out:
        if key[0] == "f32":
            continue
        d = out[key][0] - ref
        da = out[key][1] - aref
        print(f"{wname:8s} {key[0]:5s} {key[1]:7s} logits: max rel {np.abs(d).max() / np.abs(ref).max():.5f}  rms rel "
              f"{np.sqrt((d ** 2).mean()) / np.sqrt((ref ** 2).mean()):.5f}   alignment rows: max abs {np.abs(da).max():.5f}  rms "
              f"{np.sqrt((da ** 2).mean()):.6f}", flush=True)
