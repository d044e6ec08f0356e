"""Golden for rows 8..63 of the BATCH-64 workload (BASELINE configs[3]): the clips `bench.py`'s config3 leg transcribes beyond the
eight of e2e_bench_golden.json (noise, seeds 8..63, 30 s), same model / weights / generate kwargs as gen_golden_bench.py, through the
reference call of REF/transcribe.py:21-33 on the installed transformers 5.15.0 (CPU, fp32, one clip per pipeline call).

Recorded per clip: final text / word chunks and the decoder sequences of every inner generate pass (no token-timestamp dump: the
words carry them).  Clips are generated in a spread order (8, 63, 16, 24, ... first) and the file is rewritten after every clip, so
an interrupted run still leaves clips from all over the batch.

    python -m tests.golden.gen_golden_bench64 [n_threads]      (~3 CPU minutes per clip on 8 cores, ~25 GB RAM)
Writes tests/golden/e2e_bench_b64_golden.json."""
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


def seed_order():
    first = [8, 63, 16, 24, 32, 40, 48, 56, 12, 20, 28, 36, 44, 52, 60, 17]
    return first + [s for s in range(8, 64) if s not in first]


def main():
    torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count())
    g, v = syn.large_v3_geometry()
    model = H.build_model(g, v, n_align=15)
    sd = {n: torch.from_numpy(syn.weight_tensor(g, n, shape, 0, "aligned")) for n, shape in syn.weight_shapes(g).items()}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    model.load_state_dict(sd, strict=True)
    del sd
    model.generation_config.alignment_heads = syn.alignment_heads(g, 15)
    pipe = H.build_pipeline(model, H.build_tokenizer(v), H.build_feature_extractor(g), batch_size=1)
    passes = []
    orig = model._extract_token_timestamps

    def spy(generate_outputs, alignment_heads, time_precision=0.02, num_frames=None, num_input_ids=None):
        passes.append(generate_outputs["sequences"].numpy().astype(np.int64).tolist())
        return orig(generate_outputs, alignment_heads, time_precision=time_precision, num_frames=num_frames, num_input_ids=num_input_ids)

    model._extract_token_timestamps = spy
    path = os.path.join(OUT, "e2e_bench_b64_golden.json")
    meta = {"weights": "aligned", "weight_seed": 0, "generate_kwargs": GEN_KW, "clips": []}
    if os.path.exists(path):
        meta = json.load(open(path))
    done = {c["seed"] for c in meta["clips"]}
    for seed in seed_order():
        if seed in done:
            continue
        x = syn.synth_audio(seed, 480000, "noise")
        passes.clear()
        t0 = time.time()
        res = pipe(x.copy(), generate_kwargs=dict(GEN_KW))
        print("clip", seed, "%.0f s" % (time.time() - t0), len(res["chunks"]), "words", len(passes), "passes", flush=True)
        meta["clips"].append({"seed": seed, "kind": "noise", "secs": 30, "text": res["text"],
                              "chunks": [{"text": c["text"], "timestamp": list(c["timestamp"])} for c in res["chunks"]],
                              "pass_sequences": [list(p) for p in passes]})
        tmp = path + ".tmp"
        json.dump(meta, open(tmp, "w"), ensure_ascii=True, indent=0)
        os.replace(tmp, path)


if __name__ == "__main__":
    main()
