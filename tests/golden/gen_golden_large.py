"""Full-size golden: the reference pipeline call (REF/transcribe.py:21-33 + README pause split) through the installed
transformers (5.15.0, CPU, fp32) at the BASELINE geometry -- Whisper large-v3 shapes, 32 + 32 layers, vocab 51866,
15 alignment heads -- with the seeded synthetic weights of ``crisperwhisper_amd.synthetic`` (no checkpoint offline).

Run from the repo root (needs /usr/local/lib/python3.10/dist-packages/transformers, ~20 GB RAM, a few CPU minutes):
    python -m tests.golden.gen_golden_large
Writes tests/golden/e2e_large_golden.json (text, word chunks, per-call token ids and token timestamps)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from crisperwhisper_amd import synthetic as syn
from tests.golden import hf_synth as H

OUT = os.path.dirname(os.path.abspath(__file__))
SCENARIOS = {
    # name: (audio kind, seconds, audio seed, weight seed, generate kwargs)
    "large_mixed30_n32": ("mixed", 30, 21, 0, {"num_beams": 1, "language": "<|en|>", "task": "transcribe", "max_new_tokens": 32}),
    "large_noise12_n24": ("noise", 12, 22, 0, {"num_beams": 1, "language": "<|en|>", "task": "transcribe", "max_new_tokens": 24}),
    # 4 chunks (30, 30, 30, 10 s) with 5 s strides, two per batch: seam merge + batch lock-step at full size
    "large_mixed70_b2_n20": ("mixed", 70, 23, 0, {"num_beams": 1, "language": "<|en|>", "task": "transcribe", "max_new_tokens": 20}),
    # beam search at full size: 5 beams on one 30 s chunk, 3 beams on 2 chunks per batch without a language (detected)
    "large_beam5_mixed30_n16": ("mixed", 30, 24, 0, {"num_beams": 5, "language": "<|en|>", "task": "transcribe", "max_new_tokens": 16}),
    "large_beam3_noise45_b2_n12_autolang": ("noise", 45, 25, 0, {"num_beams": 3, "max_new_tokens": 12}),
}
BATCH = {"large_mixed70_b2_n20": 2, "large_beam3_noise45_b2_n12_autolang": 2}


def main():
    torch.set_num_threads(os.cpu_count())
    g, v = syn.large_v3_geometry()
    t0 = time.time()
    W = syn.random_weights(g, seed=0)
    model = H.build_model(g, v, n_align=15)
    sd = {k: torch.from_numpy(x) for k, x in W.items()}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    model.load_state_dict(sd, strict=True)
    del sd
    model.generation_config.alignment_heads = syn.alignment_heads(g, 15)
    tok = H.build_tokenizer(v)
    fe = H.build_feature_extractor(g)
    print("model ready in %.0f s" % (time.time() - t0), flush=True)
    path = os.path.join(OUT, "e2e_large_golden.json")
    meta = json.load(open(path)) if os.path.exists(path) and "--all" not in sys.argv else {}
    for name, (kind, secs, seed, wseed, gk) in SCENARIOS.items():
        if name in meta:
            meta[name].setdefault("batch_size", BATCH.get(name, 1))
            continue
        x = syn.synth_audio(seed, int(round(secs * 16000)), kind)
        pipe = H.build_pipeline(model, tok, fe, batch_size=BATCH.get(name, 1))
        calls = []
        orig = model.generate

        def spy(*a, **k):
            out = orig(*a, **k)
            calls.append(out)
            return out

        model.generate = spy
        t0 = time.time()
        try:
            res = pipe(x.copy(), generate_kwargs=dict(gk))
        finally:
            model.generate = orig
        print(name, "%.0f s" % (time.time() - t0), res["text"][:60].encode(), len(res["chunks"]), "words", len(calls), "generate calls", flush=True)
        meta[name] = {
            "kind": kind, "secs": secs, "seed": seed, "weight_seed": wseed, "generate_kwargs": gk, "batch_size": BATCH.get(name, 1),
            "text": res["text"], "chunks": [{"text": c["text"], "timestamp": list(c["timestamp"])} for c in res["chunks"]],
            "n_generate_calls": len(calls),
            "sequences": [out["sequences"].numpy().astype(np.int64).tolist() for out in calls],
            "token_timestamps": [[torch.cat([s["token_timestamps"] for s in segs]).numpy().astype(np.float64).round(4).tolist() if segs else []
                                  for segs in out["segments"]] for out in calls],
        }
    json.dump(meta, open(os.path.join(OUT, "e2e_large_golden.json"), "w"), ensure_ascii=True, indent=0)


if __name__ == "__main__":
    main()
