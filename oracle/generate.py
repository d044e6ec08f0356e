"""Greedy generation + Whisper seek loop with token timestamps, numpy restatement (test oracle).

Follows TF/models/whisper/generation_whisper.py:
  generate :383-968 (seek loop :785-903), generate_with_fallback :970-1116 (single temperature,
  no thresholds), _postprocess_outputs :1129-1192, _retrieve_segment :1977-2074,
  _maybe_reduce_batch :1814-1829, _get_input_segment :1831-1852, _retrieve_init_tokens :1455-1608
  (explicit language/task only), and the greedy branch of TF/generation/utils.py:2783-2973.
Only the configuration the reference uses is covered: num_beams=1, no temperature fallback,
condition_on_prev_tokens=False, no prompt_ids, explicit language + task.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional

import numpy as np

from . import logits as LP
from . import timestamps as TS


@dataclasses.dataclass
class GenSpec:
    eos: int
    pad: int
    sot: int
    no_timestamps: int
    lang_to_id: dict
    task_to_id: dict
    alignment_heads: list
    suppress: list
    begin_suppress: list
    max_initial_timestamp_index: Optional[int] = 50
    max_length: int = 448
    median_filter_width: int = 7

    @property
    def timestamp_begin(self):
        return self.no_timestamps + 1


def greedy(model, spec: GenSpec, enc: np.ndarray, decoder_input_ids: np.ndarray, *, begin_index: int,
           max_new_tokens: Optional[int], min_new_tokens: Optional[int]):
    """One ``super().generate`` call (:1027): returns (sequences [B, prompt+gen], weights [B,H_a,L,S]).
    L = prompt + generated - 1 rows: one attention row per decoder *input* position."""
    B, n_prompt = decoder_input_ids.shape
    max_length = (n_prompt + max_new_tokens) if max_new_tokens is not None else spec.max_length
    pspec = LP.ProcessorSpec(eos=spec.eos, no_timestamps=spec.no_timestamps, suppress=spec.suppress,
                             begin_suppress=spec.begin_suppress,
                             max_initial_timestamp_index=spec.max_initial_timestamp_index,
                             min_new_tokens=min_new_tokens or 0)
    cache = model.new_cache(enc)
    ids = decoder_input_ids.copy()
    unfinished = np.ones(B, dtype=bool)
    rows = []
    feed = ids
    while True:
        lg, cross = model.decode(feed, cache, want_heads=spec.alignment_heads)
        rows.append(cross)
        scores = LP.process(pspec, ids, lg, begin_index, n_prompt)
        nxt = scores.argmax(axis=-1)
        nxt = np.where(unfinished, nxt, spec.pad)                 # utils.py:2928-2929
        ids = np.concatenate([ids, nxt[:, None]], axis=1)
        unfinished = unfinished & (nxt != spec.eos) & (ids.shape[1] < max_length)
        if not unfinished.any():
            break
        feed = nxt[:, None]
    return ids, np.concatenate(rows, axis=2)


def _retrieve_segment(seq: np.ndarray, token_ts: np.ndarray, time_offset: float, tb: int,
                      seek_num_frames: int, idx_offset: int):
    """:1977-2074 for one sequence; returns (segments, segment_offset)."""
    is_ts = seq >= tb
    single_ending = is_ts[-2:].tolist() == [False, True]
    consec = np.where(is_ts[:-1] & is_ts[1:])[0] + 1
    segments = []
    if len(consec) > 0:
        slices = consec.tolist()
        if single_ending:
            slices.append(len(seq))
        else:
            slices[-1] += 1
        last = 0
        for i, cur in enumerate(slices):
            is_last = i == len(slices) - 1
            sl = seq[last:cur]
            start_pos = int(sl[0]) - tb
            end_pos = int(sl[-1 if (not is_last or single_ending) else -2]) - tb
            segments.append({
                "start": time_offset + start_pos * 0.02, "end": time_offset + end_pos * 0.02,
                "tokens": sl, "idxs": (idx_offset + last, idx_offset + cur),
                "token_timestamps": (token_ts[idx_offset + last: idx_offset + cur] + np.float32(time_offset)).astype(np.float32),
            })
            last = cur
        if single_ending:
            seg_off = seek_num_frames
        else:
            seg_off = (int(seq[last - 2]) - tb) * 2
    else:
        segments.append({
            "start": time_offset, "end": None, "tokens": seq, "idxs": (idx_offset, idx_offset + len(seq)),
            "token_timestamps": (token_ts[idx_offset: idx_offset + len(seq)] + np.float32(time_offset)).astype(np.float32),
        })
        seg_off = seek_num_frames
    return segments, seg_off


def detect_language(model, spec: GenSpec, input_features: np.ndarray) -> np.ndarray:
    """:1610-1673: encoder on the first 3000 frames + one decoder step on <|startoftranscript|>,
    argmax over the language ids."""
    enc = model.encode(input_features[:, :, :3000])
    cache = model.new_cache(enc)
    logits, _ = model.decode(np.full((enc.shape[0], 1), spec.sot, dtype=np.int64), cache)
    lang_ids = np.array(sorted(set(spec.lang_to_id.values())), dtype=np.int64)
    return lang_ids[np.argmax(logits[:, lang_ids], axis=-1)]


def generate(model, spec: GenSpec, input_features: np.ndarray, num_frames: np.ndarray, *, language: Optional[str],
             task: Optional[str] = "transcribe", max_new_tokens: Optional[int] = None,
             min_new_tokens: Optional[int] = None, trace: Optional[list] = None):
    """input_features [B, n_mels, 3000]; num_frames [B] = attention_mask.sum(-1) (:1694).

    Returns {"sequences" [B,Lmax] (pad-right), "token_timestamps": list of [L_i] f32 (what the
    pipeline builds at TF/pipelines/automatic_speech_recognition.py:536-540), "segments"}."""
    B, _, total = input_features.shape
    nseg = 3000
    tb = spec.timestamp_begin
    if language is None:      # :1558-1590: detected language; the task token only if a task was passed
        langs = detect_language(model, spec, input_features)
        init = np.array([[spec.sot, int(l)] + ([spec.task_to_id[task]] if task is not None else []) for l in langs],
                        dtype=np.int64)
    else:
        init = np.array([[spec.sot, spec.lang_to_id[language], spec.task_to_id[task or "transcribe"]]] * B, dtype=np.int64)
    begin_index = init.shape[1]
    seek = np.zeros(B, dtype=np.int64)
    max_frames = np.full(B, total, dtype=np.int64)
    feats = input_features
    cur_map = list(range(B))
    current_segments: List[list] = [[] for _ in range(B)]
    num_frames = np.asarray(num_frames, dtype=np.int64)
    while (seek < max_frames).any():
        # _maybe_reduce_batch :1814-1829
        keep = [k for k, prev in enumerate(cur_map) if seek[prev] < max_frames[prev]]
        feats = feats[keep]
        cur_map = [cur_map[k] for k in keep]
        time_offset = seek.astype(np.float64) * 0.02 / 2
        seek_num = np.minimum(max_frames - seek, nseg)
        seg_in = np.zeros((len(cur_map), feats.shape[1], nseg), dtype=np.float32)
        for i, prev in enumerate(cur_map):
            n = int(seek_num[prev])
            seg_in[i, :, :n] = feats[i, :, seek[prev]: seek[prev] + n]
        dec_in = init[cur_map]
        mnt = max_new_tokens
        if mnt is not None and mnt + dec_in.shape[1] > spec.max_length:
            mnt = spec.max_length - dec_in.shape[1]
        enc = model.encode(seg_in)
        seqs, weights = greedy(model, spec, enc, dec_in, begin_index=dec_in.shape[1],
                               max_new_tokens=mnt, min_new_tokens=min_new_tokens)
        nf = (num_frames - seek)[cur_map]
        token_ts = TS.extract_token_timestamps(weights, nf, dec_in.shape[1], spec.median_filter_width)
        if trace is not None:
            trace.append({"seek": seek.copy(), "map": list(cur_map), "sequences": seqs.copy(),
                          "token_timestamps": token_ts.copy(), "weights": weights})
        n_in = dec_in.shape[1]
        for i, prev in enumerate(cur_map):
            s = seqs[i, n_in:]
            if s[-1] == spec.pad:                                  # :1060-1067
                npad = int((s == spec.pad).sum())
                if spec.pad == spec.eos:
                    npad -= 1
                if npad != 0:
                    s = s[:-npad]
            if s[-1] == spec.eos:                                   # :1081-1082
                s = s[:-1]
            segs, off = _retrieve_segment(s, token_ts[i], float(time_offset[prev]), tb, int(seek_num[prev]), n_in)
            seek[prev] += off
            current_segments[prev] += segs
    seq_list = [np.concatenate([d["tokens"] for d in segs]) if segs else np.zeros(0, dtype=np.int64)
                for segs in current_segments]
    Lmax = max(len(s) for s in seq_list)
    sequences = np.full((B, Lmax), spec.pad, dtype=np.int64)
    for b, s in enumerate(seq_list):
        sequences[b, :len(s)] = s
    tts = [np.concatenate([d["token_timestamps"] for d in segs]) if segs else np.zeros(0, dtype=np.float32)
           for segs in current_segments]
    return {"sequences": sequences, "token_timestamps": tts, "segments": current_segments}
