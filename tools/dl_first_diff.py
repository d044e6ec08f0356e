"""First decoder forward whose logits differ between two engine variants (development aid for csrc/declayer.hip).
usage: python tools/dl_first_diff.py ENV_A ENV_B [rows] [steps]   e.g.  "" "CW_NO_QKV_SELF=1" 2 300"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crisperwhisper_amd import synthetic as syn
from crisperwhisper_amd.engine import Engine
envs = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[1:3]]
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 2
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 300
g, v = syn.large_v3_geometry()
g.enc_layers, g.dec_layers = 1, 2
spec = syn.model_spec(g, v, n_align=15)
spec.alignment_heads = [[l, h] for l in range(2) for h in (0, 3, 7, 19)]
W = syn.random_weights(g, seed=21)
T = 3 + steps
clips = [syn.synth_audio(700 + i, 480000 - 20000 * i, "noise") for i in range(rows)]
prompt = np.tile(np.array([[v.sot, v.lang_id("en"), v.transcribe]], np.int32), (rows, 1))
caps = []
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
    cap = eng.capture_logits(rows, steps)
    seqs, lens, _ = eng.decode(prompt, max_length=T, min_new_tokens=steps)
    eng.stop_capture()
    caps.append(cap[:steps].copy())
    eng.close()
a, b = caps
d = np.abs(a - b).reshape(steps, -1).max(1)
bad = np.nonzero(d > 0)[0]
print("steps with different logits:", len(bad), "first:", bad[:10], "max |diff| there:", d[bad[:10]])
