"""Thresholds golden at the BENCH geometry (large-v3 shapes, aligned synthetic weights): the reference pipeline call with
`logprob_threshold` / `no_speech_threshold` at temperature 0 on a 100 s recording whose windows fall on both sides of the
thresholds -- the same construction as gen_golden_thresholds.py (tiny geometry), so that the 16-bit engines' log-probability
tracking and cw_transcribe's skip logic are pinned at full size (tests/test_gpu_e2e.py).

    python -m tests.golden.gen_golden_thresholds_bench [n_threads]      (~25 CPU minutes on 8 cores, ~25 GB RAM)
Writes tests/golden/e2e_thresholds_bench_golden.json."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from crisperwhisper_amd import synthetic as syn
from tests.golden import hf_synth as H
from tests.golden import gen_golden_thresholds as T

OUT = os.path.dirname(os.path.abspath(__file__))
SECONDS = 100


def main():
    torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count())
    g, v = syn.large_v3_geometry()
    model = H.build_model(g, v, n_align=15)
    sd = {n: torch.from_numpy(syn.weight_tensor(g, n, shape, 0, "aligned")) for n, shape in syn.weight_shapes(g).items()}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    model.load_state_dict(sd, strict=True)
    del sd
    model.generation_config.alignment_heads = syn.alignment_heads(g, 15)
    tok, fe = H.build_tokenizer(v), H.build_feature_extractor(g)
    x = T.audio()[: SECONDS * 16000]
    base = {"num_beams": 1, "language": "<|en|>", "task": "transcribe", "max_new_tokens": 24, "temperature": 0.0}
    probe = []
    ref = T.run(model, tok, fe, x, {**base, "logprob_threshold": -1.0e9, "no_speech_threshold": 2.0}, probe)
    print("probe", [[(round(r["avg_logprob"], 4), round(r["no_speech_prob"], 6)) for r in c] for c in probe], flush=True)
    margin, lp_thr, ns_thr, n_skip = T.pick_thresholds([[(r["avg_logprob"], r["no_speech_prob"]) for r in c] for c in probe])
    rec = []
    gk = {**base, "logprob_threshold": lp_thr, "no_speech_threshold": ns_thr}
    res = T.run(model, tok, fe, x, gk, rec)
    got_skip = sum(r["should_skip"] for c in rec for r in c)
    print("thresholds", lp_thr, ns_thr, "relative margin %.3g" % margin, "chunks", len(rec), "passes", sum(len(c) for c in rec), "skipped", got_skip,
          "(predicted", n_skip, ") words", len(res["chunks"]), "vs", len(ref["chunks"]), "without", flush=True)
    assert got_skip == n_skip and got_skip > 0
    out = {"audio": f"tests/golden/gen_golden_thresholds.py:audio()[:{SECONDS} s]", "seconds": SECONDS, "batch_size": 1, "weights": "aligned", "weight_seed": 0,
           "generate_kwargs": gk, "probe": probe, "passes": rec, "relative_margin": margin, **res, "n_words_without_thresholds": len(ref["chunks"])}
    json.dump(out, open(os.path.join(OUT, "e2e_thresholds_bench_golden.json"), "w"), ensure_ascii=True, indent=0)


if __name__ == "__main__":
    main()
