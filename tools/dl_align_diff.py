"""Where do the alignment rows of two engine variants differ (development aid for csrc/declayer.hip; the configuration of
tests/test_gpu_e2e.py::test_persistent_decoder_layer_is_bit_identical).  usage: python tools/dl_align_diff.py ENV_A ENV_B [rows] [layers]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crisperwhisper_amd import synthetic as syn
from crisperwhisper_amd.engine import Engine
envs = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[1:3]]
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 8
layers = int(sys.argv[4]) if len(sys.argv) > 4 else 4
g, v = syn.large_v3_geometry()
g.enc_layers, g.dec_layers = 1, layers
spec = syn.model_spec(g, v, n_align=15)
spec.alignment_heads = [[l, h] for l in range(layers) for h in (0, 3, 7, 19)][:15]
W = syn.random_weights(g, seed=21)
T = g.max_target_positions - 4
clips = [syn.synth_audio(700 + i, 480000 - 20000 * i, ("noise", "chirp", "mixed")[i % 3]) for i in range(rows)]
prompt = np.tile(np.array([[v.sot, v.lang_id("en"), v.transcribe]], np.int32), (rows, 1))
out = []
for env in envs:
    os.environ.update(env)
    try:
        eng = Engine(spec, dtype="bf16", max_batch=rows)
    finally:
        for k in env:
            os.environ.pop(k, None)
    eng.load_state_dict(W)
    eng.mel(clips)
    eng.encode(list(range(rows)), [0] * rows, [3000] * rows)
    cap = eng.capture_logits(rows, 24)
    seqs, lens, _ = eng.decode(prompt, max_length=T, min_new_tokens=T - 3)
    eng.stop_capture()
    out.append((seqs[:, :T].copy(), eng.alignment(rows, T - 1).copy()))
    eng.close()
(sa, aa), (sb, ab) = out
print("sequences equal:", np.array_equal(sa, sb))
d = aa != ab
print("alignment elements differing:", int(d.sum()), "of", d.size)
if d.any():
    idx = np.argwhere(d)
    print("rows:", np.unique(idx[:, 0]), "slots:", np.unique(idx[:, 1]))
    pos = np.unique(idx[:, 2])
    print("positions:", pos[:20], "...", pos[-5:], "count", len(pos))
    print("max rel diff:", float((np.abs(aa - ab)[d] / np.abs(ab[d])).max()))
