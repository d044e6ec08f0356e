"""Host control flow of ``WhisperGenerationMixin.generate`` for the word-timestamp path.

Mirrors (call surface, argument meaning, error behaviour) the parts of
TF/models/whisper/generation_whisper.py the reference pipeline reaches with
``return_timestamps="word"`` and ``num_beams=1``: init tokens (:1455-1608, explicit language/task),
the seek loop (:785-903) with batch shrinking (:1814-1829), window slicing (:1831-1852), padding and
eos stripping (:1060-1082), ``_retrieve_segment`` (:1977-2074) and the final assembly (:936-968 as
consumed by TF/pipelines/automatic_speech_recognition.py:529-540).  All tensor work is delegated to
the device through ``Engine``; this module only decides *what* to run next from the decoded ids.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional

import numpy as np

from ._native import N_FRAMES
from .engine import Engine

TIME_PRECISION = 0.02          # seconds per encoder frame (30 / 1500)
INPUT_STRIDE = 2               # conv1.stride * conv2.stride


@dataclasses.dataclass
class Segment:
    tokens: np.ndarray             # int64
    token_timestamps: np.ndarray   # float32, absolute seconds within the 30 s chunk
    idxs: tuple


def detect_language(engine: Engine, n_items: int) -> np.ndarray:
    """``WhisperGenerationMixin.detect_language`` (:1610-1673): one decoder step on <|startoftranscript|>
    over the already-encoded windows, argmax restricted to the language tokens.  Returns [n_items] ids.
    (HF runs a second encoder pass for this; the native path reuses the encoder output of the first
    seek iteration, which covers the same 3000-frame window.)"""
    spec = engine.spec
    if not spec.lang_to_id:
        raise ValueError("Cannot detect language for an English-only checkpoint: the generation config has no `lang_to_id`.")
    prompt = np.full((n_items, 1), spec.decoder_start_token_id, dtype=np.int32)
    engine.decode(prompt, max_length=2)
    logits = engine.last_logits(n_items)
    lang_ids = np.array(sorted(set(spec.lang_to_id.values())), dtype=np.int64)
    return lang_ids[np.argmax(logits[:, lang_ids], axis=-1)]


def language_to_id(spec, language: str) -> int:
    """``language_to_id`` of ``_retrieve_init_tokens`` (:1466-1486): '<|en|>', 'en' and 'english' (any case) all work."""
    from .languages import TO_LANGUAGE_CODE
    language = language.lower()
    if language in spec.lang_to_id:
        tag = language
    elif language in TO_LANGUAGE_CODE:
        tag = f"<|{TO_LANGUAGE_CODE[language]}|>"
    elif language in TO_LANGUAGE_CODE.values():
        tag = f"<|{language}|>"
    else:
        is_code = len(language) == 2
        raise ValueError(f"Unsupported language: {language}. Language should be one of:"
                         f" {list(TO_LANGUAGE_CODE.values()) if is_code else list(TO_LANGUAGE_CODE.keys())}.")
    if tag not in spec.lang_to_id:
        raise ValueError(f"{tag} is not supported by this specific model as it is not in the "
                         "`generation_config.lang_to_id`. (You should just add it to the generation config)")
    return spec.lang_to_id[tag]


def resolve_prompt(spec, language: Optional[str], task: Optional[str]):
    """``_retrieve_init_tokens`` (:1455-1608) for ``return_timestamps=True``: the decoder prompt as a list whose slot 1
    is ``None`` when the language has to be detected per item.  ``language`` / ``task`` are the call's generate_kwargs;
    the generation config's own ``language`` / ``task`` are their defaults, and when both are unset the (deprecated but
    still shipped) ``forced_decoder_ids`` of the checkpoint seed the prompt, e.g. [[1, None], [2, <|transcribe|>]]."""
    from .languages import TASK_IDS
    task = task if task is not None else getattr(spec, "task", None)
    language = language if language is not None else getattr(spec, "language", None)
    if isinstance(language, (list, tuple)):
        raise ValueError("per-item language lists are not supported on the native path: pass one language or None")
    init: List[Optional[int]] = [spec.decoder_start_token_id]
    if task is None and language is None:
        forced = getattr(spec, "forced_decoder_ids", None)
        if forced is not None:
            forced = [list(f) for f in forced]
            if forced and forced[0][0] == 1:
                i = 1
                while forced and forced[0][0] == i:
                    init.append(forced[0][1])
                    forced = forced[1:]
                    i += 1
                if forced:
                    raise ValueError(f"You are using token ids in `forced_decoder_ids` that do not seem to correctly follow "
                                     f"the prompt pattern of Whisper. Make sure that {forced} has an entry for all "
                                     f"indices >= 1 and < {forced[0][0]}.")
    lang_undefined = len(init) <= 1 or init[1] is None
    lang_id: Optional[int] = None
    detect = False
    if language is not None:
        lang_id = language_to_id(spec, language)
    elif spec.lang_to_id and lang_undefined:
        detect = True
    if lang_id is not None or detect:
        if len(init) > 1:
            init[1] = None if detect else lang_id
        else:
            init.append(None if detect else lang_id)
    if task is not None:
        if task not in TASK_IDS or task not in spec.task_to_id:
            raise ValueError(f"The `{task}` task is not supported. The task should be one of `{TASK_IDS}`")
        init.append(spec.task_to_id[task])
    elif language is not None and spec.task_to_id:
        if not any(t in init for t in spec.task_to_id.values()):
            init.append(spec.task_to_id["transcribe"])
    if init[-1] == spec.no_timestamps_token_id:          # return_timestamps=True drops a trailing <|notimestamps|>
        init = init[:-1]
    head, rest = init[:2], [t for t in init[2:] if t is not None]
    if len(head) == 2 and head[1] is None and not detect:
        head = head[:1]                                    # a None language nobody fills in is dropped like any None
    return head + rest, detect


def init_tokens(spec, language: Optional[str], task: Optional[str], lang_id: Optional[int] = None) -> List[int]:
    """<|startoftranscript|><|lang|><|task|> (no <|notimestamps|>: return_timestamps=True); ``lang_id`` fills the
    language slot when it has to be detected."""
    toks, detect = resolve_prompt(spec, language, task)
    if detect:
        if lang_id is None:
            raise ValueError("language is None and no detected language id was supplied")
        toks = [toks[0], int(lang_id)] + toks[2:]
    return [int(t) for t in toks]


def split_segments(seq: np.ndarray, token_ts: np.ndarray, time_offset: float, timestamp_begin: int,
                   seek_num_frames: int, idx_offset: int):
    """Slice one decoded window at paired timestamp tokens; returns (segments, frames to advance)."""
    is_ts = seq >= timestamp_begin
    single_ending = len(seq) >= 2 and (not is_ts[-2]) and bool(is_ts[-1])
    if len(seq) == 1:
        single_ending = False                                   # tolist() == [True] != [False, True]
    pair_ends = np.nonzero(is_ts[:-1] & is_ts[1:])[0] + 1
    off32 = np.float32(time_offset)
    out: List[Segment] = []
    if len(pair_ends):
        cuts = pair_ends.tolist()
        if single_ending:
            cuts.append(len(seq))
        else:
            cuts[-1] += 1
        prev = 0
        for cut in cuts:
            out.append(Segment(seq[prev:cut], (token_ts[idx_offset + prev: idx_offset + cut] + off32).astype(np.float32),
                               (idx_offset + prev, idx_offset + cut)))
            prev = cut
        if single_ending:
            advance = seek_num_frames
        else:
            advance = (int(seq[prev - 2]) - timestamp_begin) * INPUT_STRIDE
    else:
        out.append(Segment(seq, (token_ts[idx_offset: idx_offset + len(seq)] + off32).astype(np.float32),
                           (idx_offset, idx_offset + len(seq))))
        advance = seek_num_frames
    return out, advance


def _topk_desc(values: np.ndarray, k: int) -> np.ndarray:
    """Indices of the k largest entries per row, largest first, ties towards the lower index (torch.topk on CPU)."""
    order = np.argsort(-values, axis=-1, kind="stable")
    return order[..., :k]


def _beam_search_native_host(engine, prompt, max_length, min_new_tokens, K, length_penalty, early_stopping):
    """The search loop over the native host half (csrc/beamhost.cpp): per decoder step one `cw_beam_step`, one
    `cw_beam_host_step` (candidates in, parent / token out) and one `cw_beam_advance`.  -> (sequences [B, max_length] int64,
    beam_indices [B, max_length - n_prompt] int32, scores [B] float32) of the best hypothesis per item."""
    import ctypes as C
    from . import _native
    lib = _native.load()
    spec = engine.spec
    pr = np.ascontiguousarray(prompt, dtype=np.int32)
    B, n_prompt = pr.shape
    st = lib.cw_beam_host_new(B, K, n_prompt, int(max_length), int(spec.vocab_size), int(spec.eos_token_id),
                              int(spec.pad_token_id or 0), float(length_penalty), 1 if early_stopping is True else 0,
                              pr.ctypes.data_as(C.c_void_p))
    if not st:
        raise ValueError(f"beam search: invalid geometry (items {B}, beams {K}, prompt {n_prompt}, max_length {max_length})")
    try:
        parent = np.empty(B * K, np.int32); token = np.empty(B * K, np.int32)
        engine.beam_begin(prompt, K, max_length, min_new_tokens)
        while True:
            vals, toks = engine.beam_step(2 * K)
            vals = np.ascontiguousarray(vals, dtype=np.float32); toks = np.ascontiguousarray(toks, dtype=np.int32)
            rc = lib.cw_beam_host_step(st, vals.ctypes.data_as(C.c_void_p), toks.ctypes.data_as(C.c_void_p),
                                       parent.ctypes.data_as(C.c_void_p), token.ctypes.data_as(C.c_void_p))
            if rc < 0:
                raise RuntimeError(f"cw_beam_host_step failed ({rc})")
            if rc == 0:
                break
            engine.beam_advance(parent, token)
        seqs = np.empty((B, max_length), np.int64); bi = np.empty((B, max_length - n_prompt), np.int32)
        sc = np.empty(B, np.float32)
        rc = lib.cw_beam_host_result(st, seqs.ctypes.data_as(C.c_void_p), bi.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise RuntimeError(f"cw_beam_host_result failed ({rc})")
        return seqs, bi, sc
    finally:
        lib.cw_beam_host_free(st)


def beam_search(engine: Engine, prompt: np.ndarray, max_length: int, min_new_tokens: int, num_beams: int,
                length_penalty: float = 1.0, early_stopping=False, native_host: Optional[bool] = None):
    """Host half of ``GenerationMixin._beam_search`` (TF/generation/utils.py:3208-3520) -- running / finished
    hypotheses, length penalty, the early-stopping heuristic (:3009-3053), all in float32 like HF -- over the device half
    (``Engine.beam_begin / beam_step / beam_advance / beam_finish``: decoder forwards on items x beams rows, log-softmax +
    logits processors, per-row best candidates, cache ancestry).  Deterministic beam search only (``do_sample=False``).

    Returns (sequences [B, n_prompt + max_generated] int64 padded with pad_token_id, beam_indices [B, max_generated]
    int32 flat row indices or -1, L = decoder input positions whose alignment rows were gathered for the returned
    sequences; ``engine.token_timestamps(B, L, n_prompt, ...)`` is valid afterwards).

    ``native_host`` (default True): the bookkeeping of every step runs in ``csrc/beamhost.cpp`` (one C call per step instead
    of ~45 numpy calls, 0.23 ms per step at 8 items x 5 beams); ``False`` keeps the numpy statement below, which
    ``tests/test_beam_host.py`` holds bit-equal to the native one on random candidate streams."""
    if early_stopping is not True and early_stopping is not False:
        # transformers also knows "never" (a third stopping heuristic, generation/utils.py:3042-3053); it is not implemented
        # here and must not be mistaken for False
        raise ValueError(f"early_stopping={early_stopping!r}: only True / False are implemented")
    spec = engine.spec
    f32 = np.float32
    prompt = np.asarray(prompt, dtype=np.int64)
    B, n_prompt = prompt.shape
    K, V = int(num_beams), spec.vocab_size
    if native_host is None or native_host:
        seqs, bi, sc = _beam_search_native_host(engine, prompt, max_length, min_new_tokens, K, length_penalty, early_stopping)
        return _beam_search_tail(engine, seqs, bi, sc, n_prompt)
    eos, pad = spec.eos_token_id, spec.pad_token_id
    keep = max(2, 1 + 1) * K                                    # beams_to_keep (:3280-3281), one eos token id
    top_mask = np.arange(keep) < K
    fill = pad if pad else eos                                   # output_fill_value (:3323)
    running_seq = np.full((B, K, max_length), fill, dtype=np.int64)
    running_seq[:, :, :n_prompt] = prompt[:, None, :]
    sequences = running_seq.copy()
    running_scores = np.zeros((B, K), f32); running_scores[:, 1:] = f32(-1e9)
    beam_scores = np.full((B, K), f32(-1e9), f32)
    is_sent_finished = np.zeros((B, K), bool)
    unsat = np.ones((B, 1), bool)
    running_bi = np.full((B, K, max_length - n_prompt), -1, dtype=np.int32)
    beam_indices = running_bi.copy()
    cur_len = n_prompt
    bidx = np.arange(B)[:, None]
    engine.beam_begin(prompt, K, max_length, min_new_tokens)
    while True:
        vals, toks = engine.beam_step(keep)                      # [B*K, keep] processed log-probs, best first
        vals = vals.reshape(B, K, keep).astype(f32)
        toks = toks.reshape(B, K, keep).astype(np.int64)
        acc = (vals + running_scores[:, :, None]).astype(f32)    # :3436
        acc = np.where(toks >= 0, acc, f32(-np.inf)).reshape(B, K * keep)
        flat = (np.arange(K)[None, :, None] * V + np.maximum(toks, 0)).reshape(B, K * keep)
        # top `keep` of the union in (value desc, flattened index asc) order == torch.topk over [B, K * V] (:3147)
        order = np.lexsort((flat, -acc), axis=-1)[:, :keep]
        topk_lp = np.take_along_axis(acc, order, axis=1)
        topk_flat = np.take_along_axis(flat, order, axis=1)
        topk_beam, topk_ids = topk_flat // V, topk_flat % V
        topk_seq = running_seq[bidx, topk_beam]                   # [B, keep, max_length]
        topk_seq[:, :, cur_len] = topk_ids
        topk_bi = running_bi[bidx, topk_beam]
        topk_bi[:, :, cur_len - n_prompt] = (topk_beam + np.arange(B)[:, None] * K).astype(np.int32)
        # d. stopping criteria: eos token, max length (:3456-3462)
        hits = (topk_ids == eos) | (cur_len + 1 >= max_length)
        # e. running beams of the next iteration (:3173-3190)
        run_lp = (topk_lp + hits.astype(f32) * f32(-1.0e9)).astype(f32)
        nxt = _topk_desc(run_lp, K)
        running_seq = np.take_along_axis(topk_seq, nxt[:, :, None], axis=1)
        running_scores = np.take_along_axis(run_lp, nxt, axis=1)
        running_bi = np.take_along_axis(topk_bi, nxt[:, :, None], axis=1)
        # f. finished hypotheses (:3192-3245)
        did_top = hits & top_mask[None, :]
        lp2 = (topk_lp / f32((cur_len + 1 - n_prompt) ** length_penalty)).astype(f32)
        full = np.all(is_sent_finished, axis=-1, keepdims=True) & (early_stopping is True)
        lp2 = (lp2 + full.astype(f32) * f32(-1.0e9)).astype(f32)
        lp2 = (lp2 + (~unsat).astype(f32) * f32(-1.0e9)).astype(f32)
        lp2 = (lp2 + (~did_top).astype(f32) * f32(-1.0e9)).astype(f32)
        m_seq = np.concatenate([sequences, topk_seq], axis=1)
        m_scores = np.concatenate([beam_scores, lp2], axis=1)
        m_bi = np.concatenate([beam_indices, topk_bi], axis=1)
        m_fin = np.concatenate([is_sent_finished, did_top], axis=1)
        sel = _topk_desc(m_scores, K)
        sequences = np.take_along_axis(m_seq, sel[:, :, None], axis=1)
        beam_scores = np.take_along_axis(m_scores, sel, axis=1)
        beam_indices = np.take_along_axis(m_bi, sel[:, :, None], axis=1)
        is_sent_finished = np.take_along_axis(m_fin, sel, axis=1)
        # g. next iteration: cache re-ordering (device), stopping condition of the search as a whole
        parent = running_bi[:, :, cur_len - n_prompt].reshape(-1)
        token = running_seq[:, :, cur_len].reshape(-1)
        cur_len += 1
        best_len = cur_len - n_prompt                              # early_stopping False / length_penalty: :3042-3053
        best_possible = (running_scores[:, :1] / f32(best_len ** length_penalty)).astype(f32)
        worst_finished = np.where(is_sent_finished, beam_scores.min(axis=1, keepdims=True), f32(-1.0e9))
        unsat = unsat & np.any(best_possible > worst_finished, axis=-1, keepdims=True)
        go_on = bool(np.any(unsat)) and (not (bool(np.all(is_sent_finished)) and early_stopping is True)) and (not bool(np.all(hits)))
        if not go_on:
            break
        engine.beam_advance(parent, token)
    return _beam_search_tail(engine, sequences[:, 0, :], beam_indices[:, 0, :], beam_scores[:, 0], n_prompt)


def _beam_search_tail(engine, seq_out, bi_out, scores, n_prompt):
    """Trim to the longest returned hypothesis and gather the alignment rows of the returned sequences."""
    max_gen = int((bi_out != -1).sum(axis=1).max())
    seq_out = seq_out[:, :n_prompt + max_gen]
    bi_out = bi_out[:, :max_gen]
    # cross-attention rows of the returned sequences: HF's unrolled beam_indices (generation_whisper.py:262-303),
    # -1 (positions after a hypothesis' eos) -> row 0
    unrolled = np.concatenate([np.repeat(bi_out[:, :1], n_prompt - 1, axis=1), bi_out], axis=1) if n_prompt > 1 else bi_out
    unrolled = np.where(unrolled == -1, 0, unrolled).astype(np.int32)
    engine.beam_finish(unrolled)
    return seq_out, bi_out, unrolled.shape[1], np.asarray(scores, dtype=np.float32)


def generate(engine: Engine, n_items: int, num_frames, *, language: Optional[str], task: Optional[str] = None,
             max_new_tokens: Optional[int] = None, min_new_tokens: Optional[int] = None,
             num_beams: Optional[int] = 1, stats: Optional[dict] = None, native: Optional[bool] = None,
             logprob_threshold: Optional[float] = None, no_speech_threshold: Optional[float] = None):
    """Transcribe the ``n_items`` 30 s feature windows resident in the engine (items 0..n-1).

    Returns {"sequences": [B, Lmax] int64 (pad-right), "token_timestamps": list of float32 arrays,
    "segments": list of list of Segment (host loop only)} -- the fields the pipeline consumes.

    ``native`` (default: whenever the engine exports it) runs the seek loop inside the library
    (``cw_transcribe``, one C call per batch); ``native=False`` runs the same control flow here, stage by stage
    over ``cw_encode`` / ``cw_decode`` / ``cw_token_timestamps`` -- the two are tested to agree exactly.

    ``logprob_threshold`` / ``no_speech_threshold``: the deterministic (temperature 0) half of HF's
    ``generate_with_fallback`` (generation_whisper.py:970-1116, ``_need_fallback`` :1243-1287): a window whose average token
    log-probability is below the first AND whose no-speech probability is above the second is skipped -- seek moves on by the
    whole window, no segment (:879-881).  Re-decoding at higher temperatures is not implemented (pipeline.py refuses it)."""
    spec = engine.spec
    if no_speech_threshold is not None and logprob_threshold is None:
        raise ValueError("no_speech_threshold needs logprob_threshold as well (generation_whisper.py:1275-1285 compares both)")
    # every argument is checked before any engine state changes (a refused call must not leave its thresholds behind)
    skip_on = logprob_threshold is not None and no_speech_threshold is not None
    if skip_on and num_beams is not None and int(num_beams) > 1:
        raise ValueError("logprob_threshold / no_speech_threshold are implemented for greedy decoding only: pass num_beams=1")
    num_beams = 1 if num_beams is None else int(num_beams)
    if num_beams < 1:
        raise ValueError(f"`num_beams` has to be an integer strictly greater than 0, but is {num_beams}")
    if num_beams * n_items > engine.max_batch:
        raise ValueError(f"beam search decodes items x beams = {num_beams * n_items} rows; the engine was created with "
                         f"max_batch = {engine.max_batch}")
    if hasattr(engine, "set_thresholds"):
        engine.set_thresholds(logprob_threshold, no_speech_threshold)
    elif logprob_threshold is not None:
        raise ValueError("this engine does not implement logprob_threshold / no_speech_threshold")
    num_frames = np.asarray(num_frames, dtype=np.int64)
    if native is None:
        native = hasattr(engine, "transcribe") and num_beams == 1
    if num_beams > 1:
        native = False                                  # the seek loop runs here, beam bookkeeping in beam_search()
    if native:
        toks, detect = resolve_prompt(spec, language, task)
        if detect and not spec.lang_to_id:
            raise ValueError("Cannot detect language for an English-only checkpoint: the generation config has no `lang_to_id`.")
        if len(toks) < 2 or len(toks) > 3:
            raise ValueError(f"unsupported decoder prompt {toks}: the native path decodes from "
                             "<|startoftranscript|><|lang|>[<|task|>]")
        lang_tok = -1 if detect else int(toks[1])
        task_tok = int(toks[2]) if len(toks) > 2 else -1
        toks, tts, n_calls = engine.transcribe(
            n_items, num_frames, sot=spec.decoder_start_token_id, language_token=lang_tok, task_token=task_tok,
            max_new_tokens=-1 if max_new_tokens is None else int(max_new_tokens), min_new_tokens=min_new_tokens or 0,
            max_length=spec.max_length, lang_ids=sorted(set(spec.lang_to_id.values())) if spec.lang_to_id else None)
        if stats is not None:
            stats["generate_calls"] = stats.get("generate_calls", 0) + n_calls
        width = max((len(s) for s in toks), default=0)
        sequences = np.full((n_items, width), spec.pad_token_id, dtype=np.int64)
        for i, s in enumerate(toks):
            sequences[i, :len(s)] = s
        return {"sequences": sequences, "token_timestamps": tts, "segments": None}
    pre_encoded = False
    _, detect = resolve_prompt(spec, language, task)
    if detect:
        # language auto-detection (the reference does not pass `language`, REF/transcribe.py:33)
        engine.encode(list(range(n_items)), np.zeros(n_items, np.int64), np.full(n_items, N_FRAMES, np.int64))
        langs = detect_language(engine, n_items)
        init = np.asarray([init_tokens(spec, language, task, lang_id=l) for l in langs], dtype=np.int32)
        pre_encoded = True
    else:
        init = np.tile(np.asarray(init_tokens(spec, language, task), dtype=np.int32), (n_items, 1))
    n_prompt = init.shape[1]
    if max_new_tokens is not None and max_new_tokens + n_prompt > spec.max_target_positions:
        max_new_tokens = spec.max_target_positions - n_prompt     # :1937-1942
    tb = spec.timestamp_begin
    seek = np.zeros(n_items, dtype=np.int64)
    max_frames = np.full(n_items, N_FRAMES, dtype=np.int64)
    segments: List[List[Segment]] = [[] for _ in range(n_items)]
    n_calls = 0
    while True:
        active = [i for i in range(n_items) if seek[i] < max_frames[i]]
        if not active:
            break
        seek_num = np.minimum(max_frames - seek, N_FRAMES)
        if not (pre_encoded and n_calls == 0):            # first pass: windows already encoded for detection
            engine.encode(active, seek[active], seek_num[active])
        max_length = (n_prompt + max_new_tokens) if max_new_tokens is not None else min(spec.max_length, spec.max_target_positions)
        nsp = engine.no_speech_probs(len(active), spec.decoder_start_token_id) if skip_on else None
        if num_beams > 1:
            bs, _, L, alp = beam_search(engine, init[active], max_length, min_new_tokens or 0, num_beams)
            total = bs.shape[1]
            seqs = np.full((len(active), spec.max_target_positions), spec.pad_token_id, dtype=np.int64)
            seqs[:, :total] = bs
        else:
            seqs, lens, _ = engine.decode(init[active], max_length, min_new_tokens or 0)
            total = int(lens.max())
            L = total - 1
            alp = engine.avg_logprobs(len(active)) if skip_on else None
        token_ts = engine.token_timestamps(len(active), L, n_prompt, (num_frames - seek)[active])
        n_calls += 1
        for row, i in enumerate(active):
            if skip_on and float(alp[row]) < logprob_threshold and float(nsp[row]) > no_speech_threshold:
                seek[i] += seek_num[i]                            # should_skip (:879-881)
                continue
            s = seqs[row, n_prompt:total].astype(np.int64)
            if s[-1] == spec.pad_token_id:                        # strip right padding, keep one eos
                npad = int((s == spec.pad_token_id).sum())
                if spec.pad_token_id == spec.eos_token_id:
                    npad -= 1
                if npad:
                    s = s[:-npad]
            if s[-1] == spec.eos_token_id:
                s = s[:-1]
            segs, advance = split_segments(s, token_ts[row], float(seek[i]) * TIME_PRECISION / INPUT_STRIDE, tb,
                                           int(seek_num[i]), n_prompt)
            seek[i] += advance
            segments[i].extend(segs)
    if stats is not None:
        stats["generate_calls"] = stats.get("generate_calls", 0) + n_calls
    seq_list = [np.concatenate([s.tokens for s in segs]) if segs else np.zeros(0, np.int64) for segs in segments]
    width = max((len(s) for s in seq_list), default=0)
    sequences = np.full((n_items, width), spec.pad_token_id, dtype=np.int64)
    for i, s in enumerate(seq_list):
        sequences[i, :len(s)] = s
    tts = [np.concatenate([s.token_timestamps for s in segs]) if segs else np.zeros(0, np.float32) for segs in segments]
    return {"sequences": sequences, "token_timestamps": tts, "segments": segments}
