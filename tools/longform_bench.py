"""BASELINE configs[2] on ONE GPU: a 10-minute recording (9.6 M samples -> 30 chunks of 30 s with 5 s strides) through the
public pipeline call (REF/transcribe.py:21-33 + pause split), large-v3 geometry, bf16, synthetic weights and audio.
Everything is inside the timed call: chunking, PCM upload, mel, encoder, decoder, alignment/DTW, word collation across the
29 seams, pause split.  usage: python tools/longform_bench.py [--batch 8] [--contexts 1] [--tokens 128]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import crisperwhisper_amd as cw
from crisperwhisper_amd import collate, synthetic as syn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--contexts", type=int, default=1)
    ap.add_argument("--tokens", type=int, default=128)
    ap.add_argument("--seconds", type=int, default=600)
    a = ap.parse_args()
    g, v = syn.large_v3_geometry()
    spec = syn.model_spec(g, v, n_align=15)

    class LazyWeights(dict):
        def items(self):
            for n, shape in syn.weight_shapes(g).items():
                yield n, syn.random_tensor(g, n, shape, seed=0)
    t0 = time.perf_counter()
    pipe = cw.pipeline("automatic-speech-recognition", model=cw.ModelBundle(spec, LazyWeights()),
                       tokenizer=collate.Vocabulary.from_synthetic(v), chunk_length_s=30, batch_size=a.batch,
                       return_timestamps="word", device="cuda:0", contexts=a.contexts)
    t_load = time.perf_counter() - t0
    x = syn.synth_audio(0, a.seconds * 16000, "mixed")
    gk = {"num_beams": 1, "language": "<|en|>", "task": "transcribe", "max_new_tokens": a.tokens, "min_new_tokens": a.tokens}
    pipe(x[: 60 * 16000], generate_kwargs=gk)                     # warm-up: graphs, allocations
    pipe.stats.clear()
    t0 = time.perf_counter()
    out = cw.adjust_pauses_for_hf_pipeline_output(pipe(x, generate_kwargs=gk))
    wall = time.perf_counter() - t0
    print(json.dumps({"workload": f"{a.seconds} s recording -> {len(x) // 320000 + (1 if len(x) % 320000 else 0)} chunks, batch {a.batch}, "
                                  f"{a.contexts} context(s), {a.tokens} tokens/pass, large-v3 geometry bf16, 1 GPU",
                      "wall_s": round(wall, 3), "rtf": round(wall / a.seconds, 6), "words": len(out["chunks"]),
                      "words_per_s": round(len(out["chunks"]) / wall, 1), "seek_passes": pipe.stats.get("generate_calls"),
                      "weight_load_s": round(t_load, 1)}))


if __name__ == "__main__":
    main()
