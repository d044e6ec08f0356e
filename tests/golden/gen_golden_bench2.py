"""More goldens at the BENCH geometry (round 3; same reference call and recording as gen_golden_bench.py):

  longform   the 600 s recording of bench.py's BASELINE configs[2] leg (synth_audio(1000, 600 s, "mixed")) through
             transformers.pipeline(chunk_length_s=30, batch_size=4, return_timestamps="word"): 30 chunks with 5 s strides,
             29 seams, 128 forced-length tokens per generate call, aligned weights seed 0.  (batch 4, not the bench leg's 8:
             the eager word-timestamp path keeps 5.76 GB of encoder attention maps per chunk in flight and this box has 62 GB.)
  seed1      a second weight seed (aligned, seed 1) on other audio: 2 "mixed" + 2 "chirp" clips of 30 s, one clip per call.

    python -m tests.golden.gen_golden_bench2 [longform] [seed1]        (~30 + ~12 CPU minutes on 8 cores)
Writes tests/golden/e2e_bench_longform_golden.json and tests/golden/e2e_bench_seed1_golden.json."""
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
N_TOK = 128
GEN_KW = {"num_beams": 1, "language": "<|en|>", "task": "transcribe", "max_new_tokens": N_TOK, "min_new_tokens": N_TOK}


def build(weight_seed):
    g, v = syn.large_v3_geometry()
    t0 = time.time()
    model = H.build_model(g, v, n_align=15)
    sd = {n: torch.from_numpy(syn.weight_tensor(g, n, shape, weight_seed, "aligned")) for n, shape in syn.weight_shapes(g).items()}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    model.load_state_dict(sd, strict=True)
    del sd
    model.generation_config.alignment_heads = syn.alignment_heads(g, 15)
    print("model (weight seed %d) ready in %.0f s" % (weight_seed, time.time() - t0), flush=True)
    return model, H.build_tokenizer(v), H.build_feature_extractor(g)


def words(res):
    return [{"text": c["text"], "timestamp": list(c["timestamp"])} for c in res["chunks"]]


def longform():
    model, tok, fe = build(0)
    pipe = H.build_pipeline(model, tok, fe, batch_size=4)
    secs = 600
    x = syn.synth_audio(1000, secs * 16000, "mixed")
    t0 = time.time()
    res = pipe(x.copy(), generate_kwargs=dict(GEN_KW))
    print("longform %.0f s, %d words" % (time.time() - t0, len(res["chunks"])), flush=True)
    meta = {"weights": "aligned", "weight_seed": 0, "generate_kwargs": GEN_KW, "audio": {"seed": 1000, "kind": "mixed", "secs": secs},
            "pipeline": {"chunk_length_s": 30, "batch_size": 4}, "text": res["text"], "chunks": words(res)}
    json.dump(meta, open(os.path.join(OUT, "e2e_bench_longform_golden.json"), "w"), ensure_ascii=True, indent=0)


def seed1():
    model, tok, fe = build(1)
    pipe = H.build_pipeline(model, tok, fe, batch_size=1)
    meta = {"weights": "aligned", "weight_seed": 1, "generate_kwargs": GEN_KW, "clips": []}
    for seed, kind in ((200, "mixed"), (201, "chirp"), (202, "mixed"), (203, "chirp")):
        x = syn.synth_audio(seed, 480000, kind)
        if kind == "chirp":                       # synth chirp is deterministic: vary it per clip
            x = np.roll(x, seed * 1000) * (1.0 + 0.1 * (seed % 3))
        t0 = time.time()
        res = pipe(x.astype(np.float32).copy(), generate_kwargs=dict(GEN_KW))
        print("clip", seed, kind, "%.0f s" % (time.time() - t0), len(res["chunks"]), "words", flush=True)
        meta["clips"].append({"seed": seed, "kind": kind, "secs": 30, "text": res["text"], "chunks": words(res)})
        json.dump(meta, open(os.path.join(OUT, "e2e_bench_seed1_golden.json"), "w"), ensure_ascii=True, indent=0)


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    what = sys.argv[1:] or ["longform", "seed1"]
    if "seed1" in what:
        seed1()
    if "longform" in what:
        longform()
