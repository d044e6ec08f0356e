"""Beam-search goldens: the reference pipeline call through the installed transformers 5.15.0 (CPU, fp32, tiny
synthetic model) with ``num_beams`` > 1 -- including the LITERAL call of REF/transcribe.py:33, which passes no generate_kwargs and
therefore runs the 5.x pipeline defaults: 5 beams, language auto-detection, max_length 448
(TF/pipelines/automatic_speech_recognition.py:160-163, TF/pipelines/base.py:887-908).

    python -m tests.golden.gen_golden_beam      -> tests/golden/e2e_beam_golden.json / .npz"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from crisperwhisper_amd import synthetic as syn
from tests.golden import hf_synth as H
from tests.golden.gen_golden import build_tiny

OUT = os.path.dirname(os.path.abspath(__file__))
EN = {"language": "<|en|>", "task": "transcribe"}
SCENARIOS = {
    # name: (audio kind, seconds, seed, batch_size, generate kwargs exactly as passed)
    "beam5_mixed40_b2_n24": ("mixed", 40, 31, 2, {**EN, "num_beams": 5, "max_new_tokens": 24}),
    "beam3_noise30_b1_n40": ("noise", 30, 32, 1, {**EN, "num_beams": 3, "max_new_tokens": 40}),
    "beam2_chirp12_b1_min16": ("chirp", 12, 33, 1, {**EN, "num_beams": 2, "max_new_tokens": 16, "min_new_tokens": 16}),
    "beam5_noise50_b3_free": ("noise", 50, 34, 3, {**EN, "num_beams": 5}),
    "literal_reference_call_noise20": ("noise", 20, 35, 16, {}),          # REF/transcribe.py:33: pipe(file)
}


def main():
    torch.set_num_threads(2)
    torch.manual_seed(0)
    g, v, W, model = build_tiny()
    tok = H.build_tokenizer(v)
    fe = H.build_feature_extractor(g)
    meta, arrays = {}, {}
    for name, (kind, secs, seed, bs, gk) in SCENARIOS.items():
        x = syn.synth_audio(seed, int(round(secs * 16000)), kind)
        pipe = H.build_pipeline(model, tok, fe, batch_size=bs)
        calls = []
        orig = model.generate

        def spy(*a, **k):
            out = orig(*a, **k)
            calls.append(out)
            return out

        model.generate = spy
        try:
            res = pipe(x.copy(), generate_kwargs=dict(gk)) if gk else pipe(x.copy())
        finally:
            model.generate = orig
        meta[name] = {"kind": kind, "secs": secs, "seed": seed, "batch_size": bs, "generate_kwargs": gk, "text": res["text"],
                      "chunks": [{"text": c["text"], "timestamp": list(c["timestamp"])} for c in res["chunks"]],
                      "n_generate_calls": len(calls)}
        for ci, out in enumerate(calls):
            arrays[f"{name}/call{ci}/sequences"] = out["sequences"].numpy().astype(np.int64)
            for bi, segs in enumerate(out["segments"]):
                arrays[f"{name}/call{ci}/tts{bi}"] = (torch.cat([s["token_timestamps"] for s in segs]).numpy().astype(np.float32)
                                                      if segs else np.zeros(0, np.float32))
        print(name, len(res["chunks"]), "words", len(calls), "generate calls", res["text"][:50].encode(), flush=True)
    json.dump(meta, open(os.path.join(OUT, "e2e_beam_golden.json"), "w"), ensure_ascii=True, indent=0)
    np.savez_compressed(os.path.join(OUT, "e2e_beam_golden.npz"), **arrays)


if __name__ == "__main__":
    main()
