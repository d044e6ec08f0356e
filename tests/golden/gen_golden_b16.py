"""Golden at the reference's own default batch size (REF/transcribe.py:27: batch_size=16): a 450 s recording = 18 chunks of 30 s
(5 s strides) through transformers.pipeline(chunk_length_s=30, batch_size=16, return_timestamps="word") -- one full batch of 16
windows plus a ragged one of 2 -- on the tiny geometry (at the large-v3 geometry the eager word-timestamp path keeps 5.76 GB of
attention maps per window in flight: 92 GB at batch 16, more than this box has).  Batch composition matters to the reference:
when all windows of a generate call have the same num_frames HF crops the alignment matrix twice (generation_whisper.py:318-323,
:354), and the seek loop shrinks the batch as windows finish.

    python -m tests.golden.gen_golden_b16          (~1 CPU minute)
Writes tests/golden/e2e_b16_golden.json."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from crisperwhisper_amd import synthetic as syn
from tests.golden import hf_synth as H
from tests.golden.gen_golden import build_tiny, GEN_KW

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    g, v, W, model = build_tiny()
    tok = H.build_tokenizer(v)
    fe = H.build_feature_extractor(g)
    out = {}
    for name, (kind, secs, seed, extra) in {"mixed450_b16_n24": ("mixed", 450, 12, {"max_new_tokens": 24}),
                                            "noise400_b16_free": ("noise", 400, 13, {})}.items():
        x = syn.synth_audio(seed, secs * 16000, kind)
        pipe = H.build_pipeline(model, tok, fe, batch_size=16)
        calls = []
        orig = model.generate

        def spy(*a, **k):
            calls.append(int(k["input_features"].shape[0]) if "input_features" in k else -1)
            return orig(*a, **k)

        model.generate = spy
        try:
            res = pipe(x.copy(), generate_kwargs={**GEN_KW, **extra})
        finally:
            model.generate = orig
        print(name, len(res["chunks"]), "words, generate calls with batch sizes", calls, flush=True)
        out[name] = {"kind": kind, "secs": secs, "seed": seed, "batch_size": 16, "extra": extra, "text": res["text"],
                     "chunks": [{"text": c["text"], "timestamp": list(c["timestamp"])} for c in res["chunks"]],
                     "generate_batch_sizes": calls}
    json.dump(out, open(os.path.join(OUT, "e2e_b16_golden.json"), "w"), ensure_ascii=True, indent=0)


if __name__ == "__main__":
    main()
