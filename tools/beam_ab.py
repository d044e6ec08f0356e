"""A/B of the two cross-attention decode kernels under beam search on the bench workload (8 clips x 5 hypotheses, large-v3
geometry, aligned synthetic weights): same engine, same inputs, kernel flipped in-process.  Prints per clip whether the token
sequences agree and where they first differ (with the hypothesis scores)."""
import sys
import numpy as np
from crisperwhisper_amd import _native, generation, synthetic as syn
from crisperwhisper_amd.engine import Engine

B, NB, TOK = 8, 5, 128
g, v = syn.large_v3_geometry()
spec = syn.model_spec(g, v, n_align=15)
eng = Engine(spec, dtype=sys.argv[1] if len(sys.argv) > 1 else "bf16", max_batch=B * NB)
for name, shape in syn.weight_shapes(g).items():
    eng.load_tensor(name, syn.weight_tensor(g, name, shape, 0, "aligned"))
clips = [syn.synth_audio(i, 480000, "noise") for i in range(B)]
nf = eng.upload_pcm(clips)
lib = _native.load()
outs = {}
for path in ("mfma", "valu"):
    lib.cw_test_set_option(b"cross_valu", 1 if path == "valu" else 0)
    eng.mel_resident(B)
    outs[path] = generation.generate(eng, B, nf, language="<|en|>", task="transcribe", max_new_tokens=TOK, min_new_tokens=TOK,
                                     num_beams=NB)
lib.cw_test_set_option(b"cross_valu", 0)
a, b = outs["mfma"], outs["valu"]
for k in range(B):
    sa, sb = list(a["sequences"][k]), list(b["sequences"][k])
    n = min(len(sa), len(sb))
    d = next((i for i in range(n) if sa[i] != sb[i]), None)
    ta, tb = np.asarray(a["token_timestamps"][k]), np.asarray(b["token_timestamps"][k])
    m = min(len(ta), len(tb))
    print(f"clip {k}: len {len(sa)}/{len(sb)} first_diff {d} max|dt| {np.abs(ta[:m] - tb[:m]).max() if m else 0:.3f}",
          "" if d is None else f"mfma {sa[d - 2:d + 3]} valu {sb[d - 2:d + 3]}")
