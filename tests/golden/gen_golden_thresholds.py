"""Golden for the deterministic half of generate_with_fallback (TF generation_whisper.py:970-1116, _need_fallback :1243-1287,
WhisperNoSpeechDetection logits_process.py:2050-2112): the reference pipeline call with `logprob_threshold` and
`no_speech_threshold` set at a single temperature, on a recording whose windows fall on both sides of the thresholds, so that
some windows are skipped (seek += window, no segment) and others are kept.

A first run with thresholds that can never fire records, per generate pass and row, the two quantities HF compares
(average log-probability of the generated tokens, no-speech probability at the <|startoftranscript|> position); the
thresholds are then chosen so that every chunk keeps its first pass (transformers' own pipeline cannot post-process a chunk
without segments) while later passes fall on both sides, and the run is repeated with them (greedy).

    python -m tests.golden.gen_golden_thresholds          (tiny geometry, ~1 CPU minute)
Writes tests/golden/e2e_thresholds_golden.json."""
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


def audio():
    """150 s: noise / silence / tone bursts in 10 s blocks, so that the 30 s windows (5 s strides) differ a lot."""
    rng = np.random.default_rng(77)
    n = 150 * 16000
    t = np.arange(n) / 16000.0
    x = np.zeros(n, np.float64)
    for blk in range(15):
        lo, hi = blk * 160000, (blk + 1) * 160000
        kind = blk % 3
        if kind == 0:
            x[lo:hi] = rng.standard_normal(hi - lo) * (0.02 + 0.03 * (blk % 5))
        elif kind == 1:
            x[lo:hi] = 0.25 * np.sin(2 * np.pi * (200.0 + 40.0 * blk) * t[lo:hi]) * (np.sin(2 * np.pi * 1.3 * t[lo:hi]) > 0)
        # kind 2: hard silence
    return x.astype(np.float32)


def pick_thresholds(probe):
    """probe: per chunk, the (avg_logprob, no_speech_prob) of its passes in order.  A pass is skipped when its avg_logprob is
    below the first threshold and its no-speech probability above the second; a skip ends its chunk.  transformers' own
    pipeline cannot post-process a chunk without segments (torch.cat of an empty list, pipelines/automatic_speech_recognition.py
    :536-537), so the thresholds must keep the FIRST pass of every chunk and skip at least one later pass; among those the pair
    farthest (relative) from every probed value is taken."""
    lps = sorted({r[0] for c in probe for r in c})
    nss = sorted({r[1] for c in probe for r in c})
    mid = lambda v: [0.5 * (a + b) for a, b in zip(v[:-1], v[1:])]
    best = None
    for lt in mid(lps):
        for nt in mid(nss):
            skip = lambda r: r[0] < lt and r[1] > nt
            if any(skip(c[0]) for c in probe):
                continue
            n_skip = sum(any(skip(r) for r in c[1:]) for c in probe)
            n_keep_later = sum(1 for c in probe for r in c[1:] if not skip(r))
            if n_skip == 0 or n_keep_later == 0:
                continue
            margin = min(min(abs(r[0] - lt) / abs(lt), abs(r[1] - nt) / abs(nt)) for c in probe for r in c)
            if best is None or margin > best[0]:
                best = (margin, lt, nt, n_skip)
    assert best is not None, "no threshold pair keeps every first pass and skips a later one"
    return best


def run(model, tok, fe, x, gk, record):
    import transformers.models.whisper.generation_whisper as GW
    from transformers.generation.logits_process import WhisperNoSpeechDetection
    pipe = H.build_pipeline(model, tok, fe, batch_size=1)
    orig, orig_gen = model._need_fallback, model.generate

    def spy(seek_sequence, seek_outputs, index, logits_processor, generation_config, vocab_size, temperature):
        nf, sk = orig(seek_sequence, seek_outputs, index, logits_processor, generation_config, vocab_size, temperature)
        if hasattr(seek_outputs[0], "sequences_scores") or "sequences_scores" in seek_outputs[0]:
            lp = float([s["sequences_scores"] for s in seek_outputs][index])
        else:
            lp = float(model._retrieve_avg_logprobs(seek_outputs[index]["scores"], seek_sequence, temperature))
        nsp = float(GW._get_attr_from_logit_processors(logits_processor, WhisperNoSpeechDetection, "no_speech_prob")[index])
        record[-1].append({"avg_logprob": lp, "no_speech_prob": nsp, "should_skip": bool(sk), "n_tokens": int(len(seek_sequence))})
        return nf, sk

    def gen_spy(*a, **k):
        record.append([])                                   # one generate call per chunk (batch_size = 1)
        return orig_gen(*a, **k)

    model._need_fallback, model.generate = spy, gen_spy
    try:
        res = pipe(x.copy(), generate_kwargs=dict(gk))
    finally:
        model._need_fallback, model.generate = orig, orig_gen
    return {"text": res["text"], "chunks": [{"text": c["text"], "timestamp": list(c["timestamp"])} for c in res["chunks"]]}


def main():
    g, v, W, model = build_tiny()
    tok = H.build_tokenizer(v)
    fe = H.build_feature_extractor(g)
    x = audio()
    out = {"audio": "tests/golden/gen_golden_thresholds.py:audio()", "batch_size": 1, "cases": {}}
    for name, beams in (("greedy", 1),):     # beam search + thresholds: HF scores the beams differently again; refused by the native path
        # temperature must be given: transformers raises a TypeError in _retrieve_avg_logprobs when logprob_threshold is set and
        # temperature is None (generation_whisper.py:1959)
        base = {"num_beams": beams, "language": "<|en|>", "task": "transcribe", "max_new_tokens": 24, "temperature": 0.0}
        probe = []
        ref = run(model, tok, fe, x, {**base, "logprob_threshold": -1.0e9, "no_speech_threshold": 2.0}, probe)
        margin, lp_thr, ns_thr, n_skip = pick_thresholds([[(r["avg_logprob"], r["no_speech_prob"]) for r in c] for c in probe])
        rec = []
        gk = {**base, "logprob_threshold": lp_thr, "no_speech_threshold": ns_thr}
        res = run(model, tok, fe, x, gk, rec)
        got_skip = sum(r["should_skip"] for c in rec for r in c)
        print(name, "thresholds", lp_thr, ns_thr, "relative margin %.3g" % margin, "chunks", len(rec), "passes", sum(len(c) for c in rec),
              "skipped", got_skip, "(predicted", n_skip, ") words", len(res["chunks"]), "vs", len(ref["chunks"]), "without", flush=True)
        assert got_skip == n_skip and got_skip > 0 and len(res["chunks"]) < len(ref["chunks"])
        out["cases"][name] = {"generate_kwargs": gk, "probe": probe, "passes": rec, "relative_margin": margin, **res,
                              "n_words_without_thresholds": len(ref["chunks"])}
    json.dump(out, open(os.path.join(OUT, "e2e_thresholds_golden.json"), "w"), ensure_ascii=True, indent=0)


if __name__ == "__main__":
    main()
