"""Beam-search golden with the bench model: 4 of the bench clips (noise, seeds 0..3, 30 s), large-v3 geometry with the *aligned*
synthetic weights, 5 beams, 40 forced-length tokens per generate call, through the reference call of REF/transcribe.py:21-33 on the
installed transformers 5.15.0 (CPU, fp32) -- what `tests/test_gpu_e2e.py::test_bench_model_beam_search_16bit_engines` holds the
bf16 / fp16 engines against (the f32 engine's beam search is pinned by e2e_beam_golden / e2e_large_golden).

    python -m tests.golden.gen_golden_bench_beam          (~10 CPU minutes, ~25 GB RAM)
Writes tests/golden/e2e_bench_beam_golden.json."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from crisperwhisper_amd import synthetic as syn
from tests.golden import hf_synth as H

OUT = os.path.dirname(os.path.abspath(__file__))
# defaults = the committed e2e_bench_beam_golden.json; round 6 adds the bench shape itself (what `bench.py --num-beams 5` decodes):
#     CW_GOLD_CLIPS=8 CW_GOLD_TOKENS=128 CW_GOLD_NAME=e2e_bench_beam128_golden.json python -m tests.golden.gen_golden_bench_beam
N_CLIPS, N_TOK = int(os.environ.get("CW_GOLD_CLIPS", "4")), int(os.environ.get("CW_GOLD_TOKENS", "40"))
NAME = os.environ.get("CW_GOLD_NAME", "e2e_bench_beam_golden.json")
GEN_KW = {"num_beams": 5, "language": "<|en|>", "task": "transcribe", "max_new_tokens": N_TOK, "min_new_tokens": N_TOK}


def main():
    torch.set_num_threads(os.cpu_count())
    g, v = syn.large_v3_geometry()
    t0 = time.time()
    model = H.build_model(g, v, n_align=15)
    sd = {n: torch.from_numpy(syn.weight_tensor(g, n, shape, 0, "aligned")) for n, shape in syn.weight_shapes(g).items()}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    model.load_state_dict(sd, strict=True)
    del sd
    model.generation_config.alignment_heads = syn.alignment_heads(g, 15)
    tok = H.build_tokenizer(v)
    fe = H.build_feature_extractor(g)
    print("model ready in %.0f s" % (time.time() - t0), flush=True)
    pipe = H.build_pipeline(model, tok, fe, batch_size=1)
    meta = {"weights": "aligned", "weight_seed": 0, "generate_kwargs": GEN_KW, "clips": []}
    for seed in range(N_CLIPS):
        x = syn.synth_audio(seed, 480000, "noise")
        t0 = time.time()
        res = pipe(x.copy(), generate_kwargs=dict(GEN_KW))
        print("clip", seed, "%.0f s" % (time.time() - t0), len(res["chunks"]), "words", flush=True)
        meta["clips"].append({"seed": seed, "kind": "noise", "secs": 30, "text": res["text"],
                              "chunks": [{"text": c["text"], "timestamp": list(c["timestamp"])} for c in res["chunks"]]})
        json.dump(meta, open(os.path.join(OUT, NAME), "w"), ensure_ascii=True, indent=0)


if __name__ == "__main__":
    main()
