"""Golden at the BENCH shape (BASELINE configs[1]): the 8 clips `bench.py` transcribes (noise, seeds 0..7, 30 s), 128 forced-length
tokens per generate call, large-v3 geometry (32 + 32 layers) with the *aligned* synthetic weights
(crisperwhisper_amd.synthetic.aligned_tensor: seeded random tensors whose alignment heads are peaked and monotone like a
trained checkpoint's), through the reference call of REF/transcribe.py:21-33 on the installed transformers 5.15.0 (CPU, fp32).

One clip per pipeline call (the eager word-timestamp path keeps 5.76 GB of encoder attention maps per clip).
Recorded per clip: final text / word chunks, and for every inner generate pass of the seek loop the decoder sequences, the
token timestamps of `_extract_token_timestamps` (generation_whisper.py:241-381) and the `num_frames` it was called with --
what a teacher-forced run of the bf16 engine is compared against (tests/test_gpu_e2e.py).

    python -m tests.golden.gen_golden_bench          (~20 CPU minutes on 8 cores, ~25 GB RAM)
Writes tests/golden/e2e_bench_golden.json."""
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
N_CLIPS, N_TOK = 8, 128
GEN_KW = {"num_beams": 1, "language": "<|en|>", "task": "transcribe", "max_new_tokens": N_TOK, "min_new_tokens": N_TOK}


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
    passes = []
    orig = model._extract_token_timestamps

    def spy(generate_outputs, alignment_heads, time_precision=0.02, num_frames=None, num_input_ids=None):
        ts = orig(generate_outputs, alignment_heads, time_precision=time_precision, num_frames=num_frames, num_input_ids=num_input_ids)
        nf = num_frames
        if nf is not None and not isinstance(nf, int):
            nf = [int(x) for x in np.asarray(nf).reshape(-1)]
        passes.append({"sequences": generate_outputs["sequences"].numpy().astype(np.int64).tolist(),
                       "token_timestamps": ts.numpy().astype(np.float64).round(4).tolist(),
                       "num_frames": nf, "num_input_ids": int(num_input_ids)})
        return ts

    model._extract_token_timestamps = spy
    path = os.path.join(OUT, "e2e_bench_golden.json")
    meta = {"weights": "aligned", "weight_seed": 0, "generate_kwargs": GEN_KW, "clips": []}
    for seed in range(N_CLIPS):
        x = syn.synth_audio(seed, 480000, "noise")
        passes.clear()
        t0 = time.time()
        res = pipe(x.copy(), generate_kwargs=dict(GEN_KW))
        print("clip", seed, "%.0f s" % (time.time() - t0), len(res["chunks"]), "words", len(passes), "passes", flush=True)
        meta["clips"].append({"seed": seed, "kind": "noise", "secs": 30, "text": res["text"],
                              "chunks": [{"text": c["text"], "timestamp": list(c["timestamp"])} for c in res["chunks"]],
                              "passes": [dict(p) for p in passes]})
        json.dump(meta, open(path, "w"), ensure_ascii=True, indent=0)


if __name__ == "__main__":
    main()
