"""CPU tests of the product's host-side logic (no GPU): chunk scheduling, seek-loop control flow,
word collation, record packing, C-ABI surface.  Device arithmetic is stood in by an oracle-backed
engine double (tests/helpers.py) so that only the *host* code under test is the product's."""
import os
import re

import numpy as np
import pytest

from crisperwhisper_amd import _native, audio, collate, dist, generation, synthetic as syn
from oracle import collate as OC
from oracle import pipeline as OPIPE
from tests import helpers as Hh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "crisperwhisper.h")).read()
    declared = sorted(set(re.findall(r"\b(cw_[a-z_0-9]+)\s*\(", hdr)))
    lib = _native.load()                      # loads without a GPU: no compute call is made here
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/crisperwhisper.h but not exported"
    assert sorted(_native.exported_symbols()) == declared
    assert lib.cw_abi_version() == 1


def w_eq(a, b):
    return [tuple(x) for x in a] == [tuple(x) for x in b]


def test_chunk_windows_match_oracle_and_survey():
    # 10-minute stream -> 30 chunks, hop 20 s, last is 20 s
    w = audio.chunk_windows(9_600_000, 480_000, 80_000, 80_000)
    assert len(w) == 30 and w[0][2] == (480000, 0, 80000) and w[-1][1] == 320000 and w[-1][2] == (320000, 80000, 0)
    for n in (1, 79_999, 80_001, 480_000, 480_001, 1_120_000, 9_600_000):
        assert w_eq(audio.chunk_windows(n, 480_000, 80_000, 80_000), OPIPE.chunk_iter(n, 480_000, 80_000, 80_000))
    with pytest.raises(ValueError):
        audio.chunk_windows(100, 10, 6, 6)


def test_shard_bounds_and_records():
    assert [h - l for l, h in dist.shard_bounds(30, 8)] == [4, 4, 4, 4, 4, 4, 3, 3]
    assert dist.shard_bounds(3, 8)[3] == (3, 3)
    rec = dist.pack_record(7, np.array([1, 2, 300]), np.array([0.5, 1.25], np.float32), (30.0, 5.0, 0.0))
    idx, toks, ts, stride = dist.unpack_record(rec)
    assert idx == 7 and toks.tolist() == [1, 2, 300] and ts.tolist() == [0.5, 1.25] and stride == (30.0, 5.0, 0.0)
    words = [{"text": " h\u00e9llo", "timestamp": (0.5, 1.29)}, {"text": "\ufffdx", "timestamp": (1.29, 2.0)}, {"text": " ", "timestamp": (2.0, 2.0)}]
    assert dist.unpack_words(dist.pack_words(5, words)) == (5, words)
    assert dist.unpack_words(dist.pack_words(6, [])) == (6, [])


@pytest.mark.parametrize("name", ["mixed70_b2_n40", "noise35_b4_free", "chirp12_b1_n24", "noise40_b2_autolang", "mixed20_b1_autolang_notask", "noise30_b1_maxlen", "noise_1sample", "noise_100ms"])
def test_host_control_flow_word_for_word(name):
    """generation.generate + collate.decode_asr (product host code) over the oracle-backed engine
    reproduce the reference pipeline output word for word."""
    g, v, W, spec = Hh.tiny_setup()
    meta = Hh.gold_json("e2e_golden.json")[name]
    z = Hh.gold_npz("e2e_golden.npz")
    x = syn.synth_audio(meta["seed"], int(round(meta["secs"] * 16000)), meta["kind"])
    eng = Hh.OracleBackedEngine(g, v, W, spec)
    vocab = collate.Vocabulary.from_synthetic(v)
    windows = audio.chunk_windows(len(x), 480000, 80000, 80000)
    outputs, call = [], 0
    for b0 in range(0, len(windows), meta["batch_size"]):
        batch = windows[b0:b0 + meta["batch_size"]]
        _, nf = eng.mel([x[s:s + n] for s, n, _, _ in batch])
        kw = {"language": "<|en|>", "task": "transcribe", **meta["extra"]}
        out = generation.generate(eng, len(batch), nf, language=kw["language"], task=kw["task"],
                                  max_new_tokens=kw.get("max_new_tokens"), min_new_tokens=kw.get("min_new_tokens"))
        assert np.array_equal(out["sequences"], z[f"{name}/call{call}/sequences"])
        for k, (_, _, st, _) in enumerate(batch):
            assert np.array_equal(out["token_timestamps"][k], z[f"{name}/call{call}/tts{k}"])
            n = len(out["token_timestamps"][k])
            outputs.append({"tokens": out["sequences"][k][:n], "token_timestamps": out["token_timestamps"][k],
                            "stride": tuple(t / 16000 for t in st)})
        call += 1
    text, words = collate.decode_asr(vocab, outputs)
    assert text == meta["text"]
    ok, why = Hh.words_equal(words, meta["chunks"])
    assert ok, why
    seg = Hh.gold_json("e2e_segments_golden.json").get(name)
    if seg is not None:      # same tokens, segment-level chunks (return_timestamps=True) vs the transformers pipeline output
        text_s, chunks_s = collate.decode_asr(vocab, [{k: o[k] for k in ("tokens", "stride")} for o in outputs], return_timestamps=True)
        assert text_s == seg["text"]
        assert [(c["text"], list(c["timestamp"])) for c in chunks_s] == [(c["text"], c["timestamp"]) for c in seg["chunks"]]


def test_collation_matches_oracle_on_random_token_streams():
    """Random byte/timestamp streams with strides: product collation == oracle restatement
    (itself pinned to transformers by the e2e goldens)."""
    g, v = syn.tiny_geometry()
    pv, ov = collate.Vocabulary.from_synthetic(v), Hh.oracle_vocab(v)
    rng = np.random.default_rng(11)
    tb = v.timestamp_begin
    for trial in range(300):
        outs = []
        n_chunks = int(rng.integers(1, 4))
        for c in range(n_chunks):
            toks, t = [], 0
            for _ in range(int(rng.integers(1, 5))):
                t0 = t + int(rng.integers(0, 200)); t1 = t0 + int(rng.integers(1, 300)); t = min(t1, 1500)
                body = rng.choice([32, 32, 46, 44, 39, 40, 65, 66, 97, 98, 99, 0xc3, 0xa9, 0xe2, 0x82, 0xac, 0xed, 0xa0, 0x80, 0xf0,
                                   0x90, 0xf4, 0x8f, 0xc0, 0xff, 0xe0, 0x9f, 0x85, 0x0b, 0x1f], size=int(rng.integers(1, 14))).tolist()
                toks += [tb + min(t0, 1500)] + body + [tb + t]
                if rng.random() < 0.3:
                    toks = toks[:-1]                       # unterminated segment
            ts = np.round(np.sort(rng.random(len(toks)) * 30), 2).astype(np.float32)
            sl = 0.0 if c == 0 else 5.0
            sr = 0.0 if c == n_chunks - 1 else 5.0
            outs.append({"tokens": np.array(toks), "token_timestamps": ts, "stride": (30.0, sl, sr)})
        a = collate.decode_asr(pv, [dict(o) for o in outs])
        b = OC.decode_asr(ov, [dict(o) for o in outs])
        assert a[0] == b[0], trial
        ok, why = Hh.words_equal(a[1], b[1])
        assert ok, (trial, why)


def test_init_tokens_errors_mirror_reference():
    g, v, W, spec = Hh.tiny_setup()
    assert generation.init_tokens(spec, "<|en|>", "transcribe") == [v.sot, v.lang_id("en"), v.transcribe]
    assert generation.init_tokens(spec, "de", None) == [v.sot, v.lang_id("de"), v.transcribe]
    with pytest.raises(ValueError):
        generation.init_tokens(spec, "<|xx|>", "transcribe")
    with pytest.raises(ValueError):
        generation.init_tokens(spec, "<|en|>", "summarize")


def test_language_names_and_case_like_reference():
    g, v, W, spec = Hh.tiny_setup()
    for name in ("English", "english", "en", "EN", "<|en|>"):
        assert generation.init_tokens(spec, name, None) == [v.sot, v.lang_id("en"), v.transcribe]
    assert generation.init_tokens(spec, "Castilian", "translate") == [v.sot, v.lang_id("es"), v.translate]
    with pytest.raises(ValueError, match="Unsupported language"):
        generation.init_tokens(spec, "klingon", None)
    with pytest.raises(ValueError, match="not supported by this specific model"):
        generation.init_tokens(spec, "french", None)          # valid Whisper language, absent from this lang_to_id


def test_prompt_resolution_matches_transformers_retrieve_init_tokens():
    """``generation.resolve_prompt`` against ``WhisperGenerationMixin._retrieve_init_tokens`` (generation_whisper.py:1455-1608)
    over the checkpoint-side settings the reference never overrides (REF/transcribe.py:33 passes no generate_kwargs):
    ``forced_decoder_ids`` as shipped by the original Whisper checkpoints ([[1, None], [2, <|transcribe|>]]), a fixed
    language in them, a trailing <|notimestamps|>, generation_config.language / .task, and the call's own kwargs."""
    pytest.importorskip("transformers")
    import copy
    import torch
    from tests.golden import hf_synth as H
    g, v, W, spec0 = Hh.tiny_setup()
    model = H.build_model(g, v, n_align=3)
    detected = v.lang_id("de")
    model.detect_language = lambda **kw: torch.tensor([detected])
    forced_variants = [None, [[1, None], [2, v.transcribe]], [[1, v.lang_id("es")], [2, v.translate]],
                       [[1, None], [2, v.transcribe], [3, v.notimestamps]], [[1, v.lang_id("zh")]], [[2, v.transcribe]]]
    n = 0
    for forced in forced_variants:
        for gc_lang, gc_task in ((None, None), ("<|es|>", None), (None, "translate"), ("german", "transcribe")):
            for kw_lang, kw_task in ((None, None), ("en", None), (None, "transcribe"), ("<|zh|>", "translate")):
                gc = copy.deepcopy(model.generation_config)
                gc.forced_decoder_ids = forced
                gc.language = kw_lang if kw_lang is not None else gc_lang       # generate kwargs override the config
                gc.task = kw_task if kw_task is not None else gc_task
                gc.return_timestamps = True
                spec = copy.deepcopy(spec0)
                spec.forced_decoder_ids, spec.language, spec.task = forced, gc_lang, gc_task
                try:
                    want = model._retrieve_init_tokens(torch.zeros(1, g.n_mels, 3000), batch_size=1, generation_config=gc,
                                                       config=model.config, num_segment_frames=3000, kwargs={})[0].tolist()
                except ValueError:
                    with pytest.raises(ValueError):
                        generation.init_tokens(spec, kw_lang, kw_task, lang_id=detected)
                    continue
                got = generation.init_tokens(spec, kw_lang, kw_task, lang_id=detected)
                assert got == want, (forced, gc_lang, gc_task, kw_lang, kw_task, got, want)
                n += 1
    assert n >= 80


def test_drop_in_objects_from_transformers():
    """The boundary accepts the reference's own objects (REF/transcribe.py:14-31): a WhisperForConditionalGeneration
    and a WhisperTokenizer are converted to the native ModelSpec / weights / Vocabulary without loss."""
    pytest.importorskip("transformers")
    from crisperwhisper_amd.pipeline import ModelBundle
    from tests.golden import hf_synth as H
    g, v, W, spec = Hh.tiny_setup()
    model = H.build_model(g, v, n_align=3)
    model.generation_config.alignment_heads = syn.alignment_heads(g, 3)
    b = ModelBundle.from_hf(model)
    for f in ("d_model", "n_heads", "ffn_dim", "enc_layers", "dec_layers", "n_mels", "vocab_size", "max_target_positions",
              "median_filter_width", "eos_token_id", "pad_token_id", "decoder_start_token_id", "no_timestamps_token_id",
              "max_initial_timestamp_index", "lang_to_id", "task_to_id", "max_length"):
        assert getattr(b.spec, f) == getattr(spec, f), f
    assert [list(h) for h in b.spec.alignment_heads] == [list(h) for h in spec.alignment_heads]
    assert list(b.spec.suppress_tokens) == list(spec.suppress_tokens) and list(b.spec.begin_suppress_tokens) == list(spec.begin_suppress_tokens)
    assert set(b.weights) == set(syn.weight_shapes(g)) and all(b.weights[k].shape == s for k, s in syn.weight_shapes(g).items())
    tok = H.build_tokenizer(v)
    hv, sv = collate.Vocabulary.from_hf_tokenizer(tok), collate.Vocabulary.from_synthetic(v)
    assert hv.token_bytes == sv.token_bytes and hv.specials == sv.specials
    assert (hv.eos, hv.timestamp_begin, hv.startofprev, hv.sot) == (sv.eos, sv.timestamp_begin, sv.startofprev, sv.sot)
    # same collation through either table
    outs = [{"tokens": np.array([v.timestamp_begin, 72, 105, 32, 0xc3, 0xa9, 46, v.timestamp_begin + 100]),
             "token_timestamps": np.linspace(0, 2, 8).astype(np.float32), "stride": (30.0, 0.0, 0.0)}]
    assert collate.decode_asr(hv, outs) == collate.decode_asr(sv, outs)


def test_pipeline_argument_errors_mirror_reference():
    import crisperwhisper_amd as cw
    with pytest.raises(KeyError):
        cw.pipeline("text-generation", model=object())
    with pytest.raises(ValueError):
        cw.pipeline("automatic-speech-recognition")
    from crisperwhisper_amd.pipeline import _device_index, _dtype_name
    with pytest.raises(ValueError):
        _device_index("cpu")
    assert _device_index("cuda:3") == 3 and _device_index(None) == 0 and _dtype_name("torch.float16") == "f16" and _dtype_name("torch.bfloat16") == "bf16" and _dtype_name("torch.float32") == "f32"


def test_timestamp_accuracy_harness():
    from crisperwhisper_amd import metrics
    ref = [{"text": " a", "timestamp": (0.0, 0.5)}, {"text": " b", "timestamp": (0.6, 1.0)}, {"text": " c", "timestamp": (2.0, 2.5)}]
    assert metrics.boundary_f1(ref, ref, 0.2) == (1.0, 1.0, 1.0) and metrics.mean_iou(ref, ref) == 1.0
    hyp = [{"text": " a", "timestamp": (0.1, 0.55)}, {"text": " b", "timestamp": (0.6, 1.5)}, {"text": " x", "timestamp": (5.0, 5.5)}]
    p, r, f = metrics.boundary_f1(ref, hyp, 0.2)
    assert p == r == 0.5 and abs(f - 0.5) < 1e-12
    assert 0.0 < metrics.mean_iou(ref, hyp) < 1.0


def test_missing_native_library_fails_loudly(monkeypatch):
    """No CPU fallback: without libcrisperwhisper.so every entry point raises NativeLibraryError."""
    from crisperwhisper_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libcrisperwhisper.so")
    with pytest.raises(_native.NativeLibraryError, match="no CPU fallback"):
        _native.load()
    g, v, W, spec = Hh.tiny_setup()
    from crisperwhisper_amd.engine import Engine
    with pytest.raises(_native.NativeLibraryError):
        Engine(spec, dtype="f32", max_batch=1)
    with pytest.raises(_native.NativeLibraryError):
        collate.decode_asr(collate.Vocabulary.from_synthetic(v), [])


def test_checkpoint_directory_loads_without_transformers(tmp_path):
    """ModelBundle.from_pretrained / Vocabulary.from_pretrained read config.json, generation_config.json, model.safetensors
    and tokenizer.json (or vocab.json + added_tokens.json) themselves; the result equals what the transformers objects give
    (REF/transcribe.py:14-19 replaced for a local snapshot).  The checkpoint is written by transformers' save_pretrained."""
    pytest.importorskip("transformers")
    import dataclasses
    import json
    import crisperwhisper_amd as cw
    from crisperwhisper_amd.languages import LANGUAGES
    from transformers.models.whisper.tokenization_whisper import LANGUAGES as HF_LANGUAGES
    from tests.golden import hf_synth as H
    assert LANGUAGES == dict(HF_LANGUAGES)
    g, v = syn.tiny_geometry()
    model = H.build_model(g, v, n_align=3)
    tok, fe = H.build_tokenizer(v), H.build_feature_extractor(g)
    d = str(tmp_path / "ckpt")
    model.save_pretrained(d); tok.save_pretrained(d); fe.save_pretrained(d)
    a, b = cw.ModelBundle.from_hf(model), cw.ModelBundle.from_pretrained(d)
    assert dataclasses.asdict(a.spec) == dataclasses.asdict(b.spec)
    wb = dict(b.weights.items())
    assert set(wb) == set(a.weights) and all(np.array_equal(a.weights[k], wb[k]) for k in a.weights)
    va, vb = collate.Vocabulary.from_hf_tokenizer(tok), collate.Vocabulary.from_pretrained(d)
    key = lambda x: (x.token_bytes, x.specials, x.eos, x.timestamp_begin, x.startofprev, x.sot)
    assert key(va) == key(vb)
    # older snapshots: vocab.json + added_tokens.json instead of tokenizer.json
    tj = json.load(open(f"{d}/tokenizer.json"))
    d2 = tmp_path / "old"; d2.mkdir()
    json.dump(tj["model"]["vocab"], open(d2 / "vocab.json", "w"))
    json.dump({t["content"]: t["id"] for t in tj["added_tokens"]}, open(d2 / "added_tokens.json", "w"))
    assert key(collate.Vocabulary.from_pretrained(str(d2))) == key(va)
    # reference-style errors
    gcfg = json.load(open(f"{d}/generation_config.json")); gcfg.pop("alignment_heads")
    json.dump(gcfg, open(f"{d}/generation_config.json", "w"))
    with pytest.raises(ValueError, match="alignment_heads"):
        cw.ModelBundle.from_pretrained(d)
    with pytest.raises(FileNotFoundError):
        collate.Vocabulary.from_pretrained(str(tmp_path))


def test_ctranslate2_directory_loads_like_the_transformers_checkpoint(tmp_path):
    """SURVEY 8(f).4, weight-layout half: a faster-whisper style directory (model.bin in CTranslate2's layout, its config.json,
    tokenizer.json) gives the same ModelSpec fields and -- for float32 storage -- bit-identical weights as the transformers
    checkpoint it was converted from; float16 and int8 storage de-quantise to within their resolution.  The writer is
    independent test code (tests/ct2_writer.py); the real converter is not available offline, so this pins the reader to the
    published layout, not to CTranslate2 itself (DESIGN 6c says so)."""
    pytest.importorskip("transformers")
    import dataclasses
    import json
    import crisperwhisper_amd as cw
    from tests import ct2_writer
    from tests.golden import hf_synth as H
    g, v = syn.tiny_geometry()
    model = H.build_model(g, v, n_align=3)
    tok = H.build_tokenizer(v)
    a = cw.ModelBundle.from_hf(model)
    for storage, tol in (("float32", 0.0), ("float16", 1e-3), ("int8", 1.2e-2)):
        d = tmp_path / storage
        d.mkdir()
        tok.save_pretrained(str(d))
        gc = model.generation_config
        json.dump({"alignment_heads": [list(h) for h in gc.alignment_heads], "suppress_ids": list(gc.suppress_tokens or []),
                   "suppress_ids_begin": list(gc.begin_suppress_tokens or []), "lang_ids": sorted(gc.lang_to_id.values())},
                  open(d / "config.json", "w"))
        ct2_writer.write_model_bin(str(d / "model.bin"), a.weights, a.spec.n_heads, a.spec.enc_layers, a.spec.dec_layers, storage)
        b = cw.ModelBundle.from_ctranslate2(str(d))
        sa, sb = dataclasses.asdict(a.spec), dataclasses.asdict(b.spec)
        for k in ("d_model", "n_heads", "ffn_dim", "enc_layers", "dec_layers", "n_mels", "vocab_size", "max_target_positions",
                  "alignment_heads", "eos_token_id", "pad_token_id", "decoder_start_token_id", "no_timestamps_token_id",
                  "suppress_tokens", "begin_suppress_tokens", "lang_to_id", "task_to_id", "median_filter_width"):
            assert sa[k] == sb[k], (storage, k, sa[k], sb[k])
        assert set(b.weights) == set(a.weights), set(a.weights) ^ set(b.weights)
        for k, wa in a.weights.items():
            wb = b.weights[k]
            assert wb.dtype == np.float32 and wb.shape == wa.shape, k
            err = float(np.abs(wa - wb).max() / (np.abs(wa).max() + 1e-12))
            assert err <= tol, (storage, k, err)
        key = lambda x: (x.token_bytes, x.specials, x.eos, x.timestamp_begin, x.startofprev, x.sot)
        assert key(collate.Vocabulary.from_pretrained(str(d))) == key(collate.Vocabulary.from_hf_tokenizer(tok))
    (tmp_path / "float32" / "config.json").write_text("{}")
    with pytest.raises(ValueError, match="alignment_heads"):
        cw.ModelBundle.from_ctranslate2(str(tmp_path / "float32"))
    open(tmp_path / "float32" / "model.bin", "wb").write(b"\x06\x00\x00\x00\x05\x00abc")
    with pytest.raises(ValueError, match="truncated"):
        cw.ModelBundle.from_ctranslate2(str(tmp_path / "float32"))


def test_output_writers_vs_reference_vtt_and_formats():
    """`writers.timestamps_to_vtt` byte for byte against REF/app.py:74-82 (executed by the golden
    generator, tests/golden/gen_golden.py:gen_vtt), incl. minute / hour carries and the %06.3f rounding; SRT and JSON
    writers: structure, numbering, comma decimals, lossless round trip."""
    import json as _json
    import re
    from crisperwhisper_amd import writers
    cases = Hh.gold_json("vtt_golden.json")
    assert len(cases) >= 5
    for c in cases:
        chunks = [{"text": w["text"], "timestamp": tuple(w["timestamp"])} for w in c["chunks"]]
        assert writers.timestamps_to_vtt(chunks) == c["vtt"]
        srt = writers.timestamps_to_srt(chunks)
        blocks = [b for b in srt.split("\n\n") if b.strip()]
        assert len(blocks) == len(chunks)
        for i, (b, w) in enumerate(zip(blocks, chunks), 1):
            lines = b.strip("\n").split("\n")
            assert lines[0] == str(i)
            assert re.fullmatch(r"\d+:\d\d:\d\d,\d\d\d --> \d+:\d\d:\d\d,\d\d\d", lines[1]), lines[1]
            assert lines[2] == w["text"].strip()
        res = {"text": "".join(w["text"] for w in chunks), "chunks": chunks}
        back = _json.loads(writers.to_json(res))
        assert back["text"] == res["text"]
        assert [(w["text"], tuple(w["timestamp"])) for w in back["chunks"]] == [(w["text"], w["timestamp"]) for w in chunks]


def test_reference_cpu_leg_of_bench_runs_the_real_pipeline():
    """bench.py's cpu_baseline (kind="reference") = oracle/hf_reference.py: transformers' pipeline called as
    REF/transcribe.py:21-33 + pause split, in a subprocess.  Tiny geometry here: the leg must produce the same words as the
    committed transformers golden pipeline would for that clip shape, report its stage split, and honour the timeout."""
    pytest.importorskip("transformers")
    import bench
    r = bench.cpu_reference("tiny", 8, 2, 300)
    assert r["kind"] == "reference" and r["value"] and r["value"] > 0, r
    assert r["cores"] == 2 and "transformers.pipeline" in r["sample"]
    slow = bench.cpu_reference("tiny", 8, 2, 1)
    assert slow["value"] is None and "did not finish" in slow["sample"]


BEAM_SCENARIOS = ["beam5_mixed40_b2_n24", "beam3_noise30_b1_n40", "beam2_chirp12_b1_min16", "beam5_noise50_b3_free",
                  "literal_reference_call_noise20"]


@pytest.mark.parametrize("name", BEAM_SCENARIOS)
def test_beam_search_host_half_word_for_word_vs_transformers(name):
    """`generation.beam_search` (numpy port of GenerationMixin._beam_search's hypothesis bookkeeping) +
    the seek loop + beam-index gathering of the alignment rows, over the oracle-backed engine, against the transformers
    pipeline run with num_beams = 2 / 3 / 5 -- and against the LITERAL reference call (no generate_kwargs: 5 beams,
    detected language): identical sequences, token timestamps, text and words."""
    g, v, W, spec = Hh.tiny_setup()
    meta = Hh.gold_json("e2e_beam_golden.json")[name]
    z = Hh.gold_npz("e2e_beam_golden.npz")
    x = syn.synth_audio(meta["seed"], int(round(meta["secs"] * 16000)), meta["kind"])
    eng = Hh.OracleBackedEngine(g, v, W, spec)
    vocab = collate.Vocabulary.from_synthetic(v)
    windows = audio.chunk_windows(len(x), 480000, 80000, 80000)
    gk = meta["generate_kwargs"]
    outputs, call = [], 0
    for b0 in range(0, len(windows), meta["batch_size"]):
        batch = windows[b0:b0 + meta["batch_size"]]
        _, nf = eng.mel([x[s:s + n] for s, n, _, _ in batch])
        out = generation.generate(eng, len(batch), nf, language=gk.get("language"), task=gk.get("task"),
                                  max_new_tokens=gk.get("max_new_tokens"), min_new_tokens=gk.get("min_new_tokens"),
                                  num_beams=gk.get("num_beams", 5))            # 5 = the installed pipeline's default
        assert np.array_equal(out["sequences"], z[f"{name}/call{call}/sequences"])
        for k, (_, _, st, _) in enumerate(batch):
            assert np.allclose(out["token_timestamps"][k], z[f"{name}/call{call}/tts{k}"], atol=1e-6)
            n = len(out["token_timestamps"][k])
            outputs.append({"tokens": out["sequences"][k][:n], "token_timestamps": out["token_timestamps"][k],
                            "stride": tuple(t / 16000 for t in st)})
        call += 1
    text, words = collate.decode_asr(vocab, outputs)
    assert text == meta["text"]
    ok, why = Hh.words_equal(words, meta["chunks"])
    assert ok, why


def test_generate_kwargs_are_honoured_or_refused():
    """pipeline._check_generate_kwargs: nothing the native path does not implement may be dropped silently (do_sample=True used
    to slip through a `val not in (None, False, 0, 0.0, 1, 1.0)` test because True == 1)."""
    from crisperwhisper_amd.pipeline import _check_generate_kwargs as chk
    for ok in ({}, {"num_beams": 5, "language": "<|de|>", "task": "translate", "max_new_tokens": 7, "min_new_tokens": 0},
               {"do_sample": False, "temperature": 0.0}, {"temperature": 0}, {"temperature": (0.0, 0.2, 0.4)},
               {"temperature": 0.0, "logprob_threshold": -1.0, "no_speech_threshold": 0.6, "compression_ratio_threshold": 1.35},
               {"num_return_sequences": 1, "prompt_ids": None, "return_timestamps": True}):
        chk(dict(ok))
    for bad in ({"do_sample": True}, {"do_sample": 1}, {"temperature": 0.5}, {"temperature": 1}, {"temperature": 1.0}, {"temperature": (1.0,)},
                {"temperature": True}, {"temperature": ()},
                {"temperature": (0.0, 0.2), "no_speech_threshold": 0.6, "logprob_threshold": -1.0},
                {"temperature": (0.0, 0.2), "compression_ratio_threshold": 1.35}, {"num_return_sequences": 2},
                {"prompt_ids": np.array([1, 2, 3])}, {"assistant_model": object()}, {"repetition_penalty": 1.1},
                {"no_repeat_ngram_size": 2}, {"condition_on_prev_tokens": True}, {"max_length": 100},
                {"no_speech_threshold": 0.6, "temperature": 0.0}, {"logprob_threshold": -1.0}, {"return_timestamps": False}):
        with pytest.raises(ValueError):
            chk(dict(bad))
    # refusals that depend on the beam width are raised up front with the width the call would actually decode with (the
    # pipeline default when the call names none), not after the audio was loaded
    thr = {"temperature": 0.0, "logprob_threshold": -1.0, "no_speech_threshold": 0.6}
    with pytest.raises(ValueError, match="pipeline default"):
        chk(dict(thr), 5)
    chk(dict(thr, num_beams=1), 5)
    chk(dict(thr), 1)


def test_pipeline_default_num_beams_wins_over_the_checkpoints_generation_config():
    """What `pipeline(...)(x)` without generate_kwargs decodes with (REF/transcribe.py:33): under the installed transformers the
    ASR pipeline's own default generation config (num_beams = 5, TF/pipelines/automatic_speech_recognition.py:160-163) is what
    `pipe.generation_config` ends up with -- also when the checkpoint's generation_config says num_beams = 1 or 3 (probed here on
    a tiny model; TF/pipelines/base.py:887-908 resolves through model._prepare_generation_config).  The drop-in's default
    (pipeline.DEFAULT_NUM_BEAMS) is therefore a constant, not read from the checkpoint."""
    transformers = pytest.importorskip("transformers")
    import importlib
    from crisperwhisper_amd import synthetic as syn
    P = importlib.import_module("crisperwhisper_amd.pipeline")
    from tests.golden import hf_synth as H
    g, v = syn.tiny_geometry()
    for nb in (None, 1, 3):
        model = H.build_model(g, v, n_align=4)
        if nb is not None:
            model.generation_config.num_beams = nb
        pipe = H.build_pipeline(model, H.build_tokenizer(v), H.build_feature_extractor(g), batch_size=1)
        assert pipe.generation_config.num_beams == P.DEFAULT_NUM_BEAMS == 5, (nb, pipe.generation_config.num_beams)


def test_bench_longform_parity_measure_and_goldens_describe_the_bench_workload():
    """bench.py's configs[2] comparison (`longform.parity` of the driver line): longest common word subsequence against the committed
    transformers output, words of it within 20 ms, identical text -- on the golden itself (word for word), with one word replaced (a
    divergence must not shift what follows it) and with a timestamp moved by 40 ms; and the goldens the bench legs load describe the
    workloads they are compared with (same recording, aligned weights, token count; 8 beam clips at 5 beams x 128 tokens)."""
    import copy
    import json
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = json.load(open(os.path.join(root, "tests", "golden", "e2e_bench_longform_golden.json")))
    assert gold["audio"] == {"seed": 1000, "kind": "mixed", "secs": 600} and gold["weights"] == "aligned"
    same = bench.longform_parity({"text": gold["text"], "chunks": gold["chunks"]}, gold)
    n = len(gold["chunks"])
    assert same["word_for_word"] and same["ok"] and same["words_within_20ms"] == n == same["reference_words"]
    other = copy.deepcopy(gold["chunks"])
    other[100]["text"] = other[100]["text"] + "x"                       # one divergent word
    other[500]["timestamp"] = [other[500]["timestamp"][0] + 0.04, other[500]["timestamp"][1]]
    r = bench.longform_parity({"text": gold["text"].replace(gold["chunks"][100]["text"], other[100]["text"], 1), "chunks": other}, gold)
    assert r["words_in_common_order"] == n - 1 and r["words_within_20ms"] == n - 2 and not r["word_for_word"] and r["ok"]
    short = bench.longform_parity({"text": "", "chunks": other[: n // 2]}, gold)
    assert not short["ok"]
    beam = bench.load_beam_goldens(128, "aligned", 5)
    assert sorted(beam) == list(range(8)) and all(len(c["chunks"]) > 50 for c in beam.values())
    assert bench.load_beam_goldens(64, "aligned", 5) == {} and bench.load_beam_goldens(128, "aligned", 3) == {}
    greedy = bench.load_bench_goldens(128, "aligned")
    assert sorted(greedy) == list(range(64))
