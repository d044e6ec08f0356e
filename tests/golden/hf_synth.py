"""HF-object builders for the synthetic oracle.

Builds, fully offline, the objects ``REF/transcribe.py:14-31`` would obtain from the hub:
a byte-level ``WhisperTokenizer``, a seeded random ``WhisperForConditionalGeneration``,
its generation config (alignment heads, suppress lists, language/task ids) and a
``WhisperFeatureExtractor`` -- laid out exactly like ``crisperwhisper_amd.synthetic``.

Only used by the golden-fixture generator and by tests that cross-check against the
installed ``transformers`` (5.15.0).  Never imported by the product path.
"""
from __future__ import annotations

import numpy as np

from crisperwhisper_amd import synthetic as syn


def build_tokenizer(vocab: syn.SynthVocab):
    from transformers import WhisperTokenizer

    b2u = syn.bytes_to_unicode()
    table = {b2u[i]: i for i in range(256)}
    for i in range(vocab.n_extra):
        table["".join(b2u[b] for b in vocab.extra_token_bytes(i))] = 256 + i
    tok = WhisperTokenizer(vocab=table, merges=[])
    assert tok.eos_token_id == vocab.eos, (tok.eos_token_id, vocab.eos)
    tok.add_special_tokens({"additional_special_tokens": vocab.special_names()[1:]})
    tok.add_tokens([f"<|{i * 0.02:.2f}|>" for i in range(1501)])
    tok.pad_token = "<|endoftext|>"
    assert len(tok) == vocab.size
    assert tok.convert_tokens_to_ids("<|notimestamps|>") == vocab.notimestamps
    assert tok.convert_tokens_to_ids("<|0.00|>") == vocab.timestamp_begin
    return tok


def build_model(geom: syn.Geometry, vocab: syn.SynthVocab, seed: int = 0, n_align: int = 15,
                weight_scale: float | None = None):
    import torch
    from transformers import WhisperConfig, WhisperForConditionalGeneration

    cfg = WhisperConfig(
        vocab_size=geom.vocab, num_mel_bins=geom.n_mels, d_model=geom.d_model,
        encoder_layers=geom.enc_layers, decoder_layers=geom.dec_layers,
        encoder_attention_heads=geom.heads, decoder_attention_heads=geom.heads,
        encoder_ffn_dim=geom.ffn, decoder_ffn_dim=geom.ffn,
        max_source_positions=geom.max_source_positions,
        max_target_positions=geom.max_target_positions,
        median_filter_width=geom.median_filter_width,
        pad_token_id=vocab.eos, bos_token_id=vocab.eos, eos_token_id=vocab.eos,
        decoder_start_token_id=vocab.sot,
    )
    torch.manual_seed(seed)
    model = WhisperForConditionalGeneration(cfg).eval()
    if weight_scale is not None:
        # widen the random weights so logits / attention rows are not near-uniform
        with torch.no_grad():
            for n, p in model.named_parameters():
                if p.ndim >= 2 and "embed_positions" not in n:
                    p.mul_(weight_scale)
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
    gc.pad_token_id = vocab.eos
    gc.eos_token_id = vocab.eos
    gc.bos_token_id = vocab.eos
    gc.decoder_start_token_id = vocab.sot
    gc.return_timestamps = False
    return model


def build_feature_extractor(geom: syn.Geometry):
    from transformers import WhisperFeatureExtractor

    return WhisperFeatureExtractor(feature_size=geom.n_mels)


def build_pipeline(model, tok, fe, batch_size=16, chunk_length_s=30):
    """The exact call of REF/transcribe.py:21-31 (CPU, fp32)."""
    import torch
    from transformers import pipeline

    return pipeline(
        "automatic-speech-recognition", model=model, tokenizer=tok, feature_extractor=fe,
        chunk_length_s=chunk_length_s, batch_size=batch_size, return_timestamps="word",
        torch_dtype=torch.float32, device="cpu",
    )


def state_dict_numpy(model):
    return {k: v.detach().cpu().numpy().astype(np.float32) for k, v in model.state_dict().items()
            if k != "proj_out.weight"}


synth_audio = syn.synth_audio
