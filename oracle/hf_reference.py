"""The reference itself, run here: ``transformers.pipeline("automatic-speech-recognition", ...)`` exactly as
REF/transcribe.py:14-33 calls it, on the CPU in fp32, followed by the pause split of REF/utils.py:1-29 (restated in
oracle/pauses.py -- /root/reference does not travel to the GPU box; transformers 5.15.0 is installed there too).

TEST / MEASUREMENT INFRASTRUCTURE ONLY (same rule as the rest of oracle/): used by ``bench.py``'s ``cpu_baseline`` leg
(kind = "reference") and by tests.  The HF objects are built offline with the synthetic vocabulary / geometry of
``crisperwhisper_amd.synthetic`` (builders: tests/golden/hf_synth.py) and receive the very tensors the GPU engine is
loaded with, so both sides run the same model.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Dict, Iterable, Tuple

import numpy as np


def build_model_fast(geom, vocab, tensors: Iterable[Tuple[str, np.ndarray]], n_align: int = 15):
    """WhisperForConditionalGeneration with the given tensors and the synthetic generation config, without paying
    for torch's random initialisation of 1.5 B parameters (meta-device construction + ``load_state_dict(assign=True)``)."""
    import torch
    from transformers import WhisperConfig, WhisperForConditionalGeneration

    from crisperwhisper_amd import synthetic as syn
    cfg = WhisperConfig(
        vocab_size=geom.vocab, num_mel_bins=geom.n_mels, d_model=geom.d_model,
        encoder_layers=geom.enc_layers, decoder_layers=geom.dec_layers,
        encoder_attention_heads=geom.heads, decoder_attention_heads=geom.heads,
        encoder_ffn_dim=geom.ffn, decoder_ffn_dim=geom.ffn,
        max_source_positions=geom.max_source_positions, max_target_positions=geom.max_target_positions,
        median_filter_width=geom.median_filter_width,
        pad_token_id=vocab.eos, bos_token_id=vocab.eos, eos_token_id=vocab.eos, decoder_start_token_id=vocab.sot)
    with torch.device("meta"):
        model = WhisperForConditionalGeneration(cfg)
    sd = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in tensors}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False, assign=True)
    if [m for m in missing if "proj_out" not in m] or unexpected:
        raise RuntimeError(f"state dict mismatch: missing {missing[:4]} unexpected {unexpected[:4]}")
    model.tie_weights()
    model.eval()
    gc = model.generation_config
    gc.no_timestamps_token_id = vocab.notimestamps
    gc.alignment_heads = syn.alignment_heads(geom, n_align)
    gc.lang_to_id = {f"<|{l}|>": vocab.lang_id(l) for l in syn.SYNTH_LANGS}
    gc.task_to_id = {"translate": vocab.translate, "transcribe": vocab.transcribe}
    gc.is_multilingual = True
    gc.max_length = geom.max_target_positions
    gc.prev_sot_token_id = vocab.startofprev
    gc.max_initial_timestamp_index = 50
    gc.suppress_tokens = vocab.suppress_tokens()
    gc.begin_suppress_tokens = vocab.begin_suppress_tokens()
    gc.pad_token_id = gc.eos_token_id = gc.bos_token_id = vocab.eos
    gc.decoder_start_token_id = vocab.sot
    gc.return_timestamps = False
    return model


def cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def time_reference_pipeline(geom, vocab, tensors, audio: np.ndarray, n_tok: int, n_align: int = 15,
                            threads: int | None = None) -> Dict:
    """One clip through the reference call (REF/transcribe.py:21-33: chunk_length_s=30, return_timestamps="word";
    batch_size=1 because the eager word-timestamp path retains 5.76 GB of encoder attention maps per clip) +
    adjust_pauses_for_hf_pipeline_output, timed end to end on the host cores.  Greedy (``num_beams=1``: the 2024 reference
    behaviour; transformers 5.x would otherwise run 5 beams) with the bench's fixed token count."""
    import torch
    import transformers
    from transformers import pipeline

    from oracle import pauses as OP
    transformers.logging.set_verbosity_error()
    from tests.golden import hf_synth as H
    threads = threads or os.cpu_count() or 1
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    model = build_model_fast(geom, vocab, tensors, n_align)
    tok = H.build_tokenizer(vocab)
    fe = H.build_feature_extractor(geom)
    t_build = time.perf_counter() - t0
    pipe = pipeline("automatic-speech-recognition", model=model, tokenizer=tok, feature_extractor=fe, chunk_length_s=30,
                    batch_size=1, return_timestamps="word", torch_dtype=torch.float32, device="cpu")
    stage = {"encoder": 0.0, "decoder": 0.0, "token_timestamps": 0.0}
    calls = {"encoder": 0, "decoder": 0, "token_timestamps": 0}

    def spy(obj, name, key):
        orig = getattr(obj, name)

        def wrapped(*a, **k):
            t = time.perf_counter()
            try:
                return orig(*a, **k)
            finally:
                stage[key] += time.perf_counter() - t
                calls[key] += 1
        setattr(obj, name, wrapped)

    spy(model.model.encoder, "forward", "encoder")
    spy(model.model.decoder, "forward", "decoder")
    spy(model, "_extract_token_timestamps", "token_timestamps")
    gk = {"num_beams": 1, "language": "<|en|>", "task": "transcribe", "max_new_tokens": n_tok, "min_new_tokens": n_tok}
    t0 = time.perf_counter()
    res = pipe(audio.copy(), generate_kwargs=gk)
    res = OP.adjust_pauses_for_hf_pipeline_output(res)
    wall = time.perf_counter() - t0
    return {"wall_s": wall, "words": len(res["chunks"]), "audio_s": len(audio) / 16000.0, "threads": torch.get_num_threads(),
            "build_s": t_build, "stage_s": {k: round(v, 3) for k, v in stage.items()}, "stage_calls": calls,
            "text": res["text"], "chunks": res["chunks"]}
