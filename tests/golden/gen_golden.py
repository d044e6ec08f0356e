"""Generates the golden fixtures in tests/golden/ by running the reference's own dependency
(transformers 5.15.0, CPU, fp32) -- and REF/utils.py -- on seeded synthetic inputs.

Run in the build container (needs /root/reference + transformers):
    python -m tests.golden.gen_golden
The fixtures are what pins oracle/ (tests/test_oracle_vs_golden.py) and the HIP path
(tests/test_gpu_*.py) to the reference; they travel to the GPU box, this script's inputs do not.
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys

import numpy as np
import torch

from crisperwhisper_amd import synthetic as syn
from tests.golden import hf_synth as H

OUT = os.path.dirname(os.path.abspath(__file__))
GEN_KW = {"num_beams": 1, "language": "<|en|>", "task": "transcribe"}


def build_tiny(seed=0, n_align=3):
    g, v = syn.tiny_geometry()
    W = syn.random_weights(g, seed=seed)
    model = H.build_model(g, v, n_align=n_align)
    sd = {k: torch.from_numpy(x) for k, x in W.items()}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    model.load_state_dict(sd, strict=True)
    model.generation_config.alignment_heads = syn.alignment_heads(g, n_align)
    return g, v, W, model


def gen_mel():
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor(feature_size=128)
    out = {}
    for kind, n in (("noise", 480000), ("mixed", 320000), ("chirp", 480000), ("noise_short", 12345)):
        x = syn.synth_audio(1, n, kind.split("_")[0])
        r = fe(x, sampling_rate=16000, return_tensors="np", return_attention_mask=True)
        out[f"{kind}_feats_sub"] = r["input_features"][0][:, ::5].astype(np.float32)
        out[f"{kind}_nframes"] = np.int64(r["attention_mask"].sum())
        out[f"{kind}_checksum"] = np.float64(r["input_features"][0].astype(np.float64).sum())
    np.savez_compressed(os.path.join(OUT, "mel_golden.npz"), **out)


def gen_align():
    """_median_filter / _dynamic_time_warping / z-score+median+mean on small seeded cases."""
    from transformers.models.whisper.generation_whisper import _dynamic_time_warping, _median_filter
    rng = np.random.default_rng(7)
    out = {}
    cases = [(5, 9), (1, 6), (7, 1), (12, 40), (3, 2), (30, 200)]
    for ci, (N, M) in enumerate(cases):
        m = rng.standard_normal((N, M)).astype(np.float32)
        if ci % 2 == 0:
            m = np.round(m * 2) / 2            # exact ties
        ti, tj = _dynamic_time_warping(-m.astype(np.float64))
        out[f"dtw{ci}_m"] = m; out[f"dtw{ci}_ti"] = ti; out[f"dtw{ci}_tj"] = tj
    out["dtw_zero_ti"], out["dtw_zero_tj"] = _dynamic_time_warping(np.zeros((3, 4)))
    for ci, (H_, N, M, w) in enumerate([(3, 6, 50, 7), (2, 4, 3, 7), (4, 9, 20, 3), (15, 20, 300, 7)]):
        a = rng.random((1, H_, N, M)).astype(np.float32)
        a = a / a.sum(-1, keepdims=True)
        t = torch.from_numpy(a)
        std = torch.std(t, dim=-2, keepdim=True, unbiased=False)
        mean = torch.mean(t, dim=-2, keepdim=True)
        z = _median_filter((t - mean) / std, w).mean(dim=1)[0].numpy()
        out[f"am{ci}_a"] = a[0]; out[f"am{ci}_w"] = np.int64(w); out[f"am{ci}_mat"] = z
        out[f"am{ci}_med"] = _median_filter(t, w)[0].numpy()
    np.savez_compressed(os.path.join(OUT, "align_golden.npz"), **out)


def gen_pauses():
    spec = importlib.util.spec_from_file_location("ref_utils", "/root/reference/utils.py")
    ref = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref)
    rng = np.random.default_rng(3)
    cases = []
    for n in (0, 1, 2, 7, 40):
        t = np.cumsum(rng.random(2 * n) * 0.3)
        t = np.round(t, 2)
        words = [{"text": f"w{i}", "timestamp": (float(t[2 * i]), float(t[2 * i + 1]))} for i in range(n)]
        if n >= 7:
            words[3]["timestamp"] = (words[2]["timestamp"][1] - 0.05, words[3]["timestamp"][1])   # overlap: gap < 0
            words[5]["timestamp"] = (words[4]["timestamp"][1], words[5]["timestamp"][1])          # gap == 0
        for thr in (0.12, 0.3):
            inp = {"text": "x", "chunks": [dict(w) for w in words]}
            outp = ref.adjust_pauses_for_hf_pipeline_output({"text": "x", "chunks": [dict(w) for w in words]}, split_threshold=thr)
            cases.append({"thr": thr, "in": inp["chunks"], "out": outp["chunks"]})
    json.dump(cases, open(os.path.join(OUT, "pauses_golden.json"), "w"))


def gen_e2e():
    g, v, W, model = build_tiny()
    tok = H.build_tokenizer(v)
    fe = H.build_feature_extractor(g)
    scenarios = {
        # name: (audio kind, seconds, seed, batch_size, extra generate kwargs)
        "mixed70_b2_n40": ("mixed", 70, 0, 2, {"max_new_tokens": 40}),
        "noise35_b4_free": ("noise", 35, 5, 4, {}),
        "chirp12_b1_n24": ("chirp", 12, 2, 1, {"max_new_tokens": 24, "min_new_tokens": 24}),
        # language auto-detection, as the reference calls it (REF/transcribe.py:33 passes no language)
        "noise40_b2_autolang": ("noise", 40, 8, 2, {"max_new_tokens": 20, "language": None, "task": "transcribe"}),
        "mixed20_b1_autolang_notask": ("mixed", 20, 4, 1, {"max_new_tokens": 20, "language": None, "task": None}),
        # maximum decode length: 445 generated tokens fill max_target_positions (448) -> DTW over 444 rows
        "noise30_b1_maxlen": ("noise", 30, 6, 1, {"min_new_tokens": 445, "max_new_tokens": 445}),
        # ragged / degenerate inputs: a one-sample clip and a 0.1 s clip
        "noise_1sample": ("noise", 1.0 / 16000, 7, 1, {"max_new_tokens": 8}),
        "noise_100ms": ("noise", 0.1, 7, 1, {"max_new_tokens": 8}),
    }
    meta = {}
    arrays = {}
    import transformers.models.whisper.generation_whisper as GW
    for name, (kind, secs, seed, bs, extra) in scenarios.items():
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
            gk = {**GEN_KW, **extra}
            gk = {k: val for k, val in gk.items() if val is not None}     # None = do not pass (auto-detect)
            res = pipe(x.copy(), generate_kwargs=gk)
        finally:
            model.generate = orig
        meta[name] = {"kind": kind, "secs": secs, "seed": seed, "batch_size": bs, "extra": extra,
                      "text": res["text"], "chunks": [{"text": c["text"], "timestamp": list(c["timestamp"])} for c in res["chunks"]],
                      "n_generate_calls": len(calls)}
        for ci, out in enumerate(calls):
            arrays[f"{name}/call{ci}/sequences"] = out["sequences"].numpy().astype(np.int64)
            tts = [torch.cat([s["token_timestamps"] for s in segs]).numpy().astype(np.float32) if segs else np.zeros(0, np.float32)
                   for segs in out["segments"]]
            for bi, t in enumerate(tts):
                arrays[f"{name}/call{ci}/tts{bi}"] = t
    # teacher-forcing fixture: encoder output + per-step logits of the first window of scenario 1
    x = syn.synth_audio(0, 70 * 16000, "mixed")[:480000]
    feats = fe(x, sampling_rate=16000, return_tensors="pt").input_features
    ids = np.array([[v.sot, v.lang_id("en"), v.transcribe] + list(arrays["mixed70_b2_n40/call0/sequences"][0][:12])], dtype=np.int64)
    with torch.no_grad():
        enc = model.model.encoder(feats).last_hidden_state
        model.config._attn_implementation = "eager"
        o = model(input_features=feats, decoder_input_ids=torch.from_numpy(ids), output_attentions=True)
    arrays["tf/feats_sub"] = feats[0][:, ::5].numpy()
    arrays["tf/enc_sub"] = enc[0][::10].numpy()
    arrays["tf/ids"] = ids
    arrays["tf/logits"] = o.logits[0].numpy()
    heads = model.generation_config.alignment_heads
    arrays["tf/cross"] = np.stack([o.cross_attentions[l][0, h].numpy() for l, h in heads])     # [Ha, T, 1500]
    json.dump(meta, open(os.path.join(OUT, "e2e_golden.json"), "w"), ensure_ascii=True, indent=0)
    np.savez_compressed(os.path.join(OUT, "e2e_golden.npz"), **arrays)


def gen_segments():
    """return_timestamps=True (segment-level chunks): the setting REF/app.py:51-61 builds its pipeline with."""
    g, v, W, model = build_tiny()
    tok = H.build_tokenizer(v)
    fe = H.build_feature_extractor(g)
    scenarios = {
        "mixed70_b2_n40": ("mixed", 70, 0, 2, {"max_new_tokens": 40}),
        "noise35_b4_free": ("noise", 35, 5, 4, {}),
        "chirp12_b1_n24": ("chirp", 12, 2, 1, {"max_new_tokens": 24, "min_new_tokens": 24}),
    }
    meta = {}
    for name, (kind, secs, seed, bs, extra) in scenarios.items():
        x = syn.synth_audio(seed, int(round(secs * 16000)), kind)
        pipe = H.build_pipeline(model, tok, fe, batch_size=bs)
        res = pipe(x.copy(), return_timestamps=True, generate_kwargs={**GEN_KW, **extra})
        meta[name] = {"kind": kind, "secs": secs, "seed": seed, "batch_size": bs, "extra": extra, "text": res["text"],
                      "chunks": [{"text": c["text"], "timestamp": list(c["timestamp"])} for c in res["chunks"]]}
    json.dump(meta, open(os.path.join(OUT, "e2e_segments_golden.json"), "w"), ensure_ascii=True, indent=0)


def gen_vtt():
    """REF/app.py cannot be imported offline (streamlit / moviepy / torchaudio are absent), so its `timestamps_to_vtt`
    (REF/app.py:74-82) is cut out of the file with `ast` and executed on seeded word lists; the product writer
    (crisperwhisper_amd/writers.py) must reproduce the strings byte for byte."""
    import ast
    from typing import Any, Dict, List, Union
    src = open("/root/reference/app.py").read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "timestamps_to_vtt")
    ns = {"List": List, "Dict": Dict, "Union": Union, "Any": Any}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "REF/app.py", "exec"), ns)
    rng = np.random.default_rng(11)
    cases = []
    for n in (0, 1, 5, 40):
        t, words = 0.0, []
        for i in range(n):
            t0 = round(t + float(rng.random()) * 2.0, 2)
            t1 = round(t0 + float(rng.random()) * 1.5, 2)
            t = t1
            words.append({"text": [" hello", " w\u00f6rld", " [UH]", " a", " 3.5", ","][int(rng.integers(0, 6))], "timestamp": [t0, t1]})
        cases.append({"chunks": words, "vtt": ns["timestamps_to_vtt"](words)})
    # hour / minute carries and millisecond rounding of the %06.3f format
    edge = [{"text": " x", "timestamp": [59.9996, 60.0]}, {"text": " y", "timestamp": [3599.5, 3600.25]},
            {"text": " z", "timestamp": [7325.125, 7325.1251]}, {"text": " q", "timestamp": [0.0005, 0.0015]}]
    cases.append({"chunks": edge, "vtt": ns["timestamps_to_vtt"](edge)})
    json.dump(cases, open(os.path.join(OUT, "vtt_golden.json"), "w"), ensure_ascii=True, indent=0)


def main():
    if "--segments-only" in sys.argv:
        gen_segments(); print("segments ok")
        return
    if "--vtt-only" in sys.argv:
        gen_vtt(); print("vtt ok")
        return
    torch.manual_seed(0)
    gen_mel(); print("mel ok")
    gen_align(); print("align ok")
    gen_pauses(); print("pauses ok")
    gen_vtt(); print("vtt ok")
    gen_e2e(); print("e2e ok")
    gen_segments(); print("segments ok")
    for f in sorted(os.listdir(OUT)):
        if f.endswith((".npz", ".json")):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
