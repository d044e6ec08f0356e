"""Logits processors of the word-timestamp path, numpy restatement (test oracle).

Order applied by HF for this path (TF/generation/utils.py:1123-1293 builds the default list,
then appends the Whisper list built at TF/models/whisper/generation_whisper.py:1774-1812):
  MinNewTokensLength (if min_new_tokens)  TF/generation/logits_process.py:203-260
  SuppressTokensAtBegin                   :1816-1866
  SuppressTokens                          :1869-1906
  WhisperTimeStamp                        :1909-2047
"""
from __future__ import annotations

import numpy as np

NEG_INF = np.float32(-np.inf)


def log_softmax(x):
    x = x.astype(np.float32)
    m = x.max(axis=-1, keepdims=True)
    s = x - m
    return (s - np.log(np.exp(s, dtype=np.float32).sum(axis=-1, keepdims=True, dtype=np.float32))).astype(np.float32)


def logsumexp(x):
    m = x.max()
    if not np.isfinite(m):
        return np.float32(m)
    return np.float32(m + np.log(np.exp(x - m, dtype=np.float32).sum(dtype=np.float32)))


class ProcessorSpec:
    def __init__(self, *, eos, no_timestamps, suppress, begin_suppress, max_initial_timestamp_index,
                 min_new_tokens=0):
        self.eos = eos
        self.no_timestamps = no_timestamps
        self.timestamp_begin = no_timestamps + 1
        self.suppress = list(suppress or [])
        self.begin_suppress = list(begin_suppress or [])
        self.max_initial_timestamp_index = max_initial_timestamp_index
        self.min_new_tokens = min_new_tokens or 0


def _masked(spec: ProcessorSpec, input_ids: np.ndarray, scores: np.ndarray, begin_index: int, prompt_len: int) -> np.ndarray:
    """Every processor up to (not including) the timestamp-mass rule of WhisperTimeStampLogitsProcessor (:2041-2045)."""
    s = scores.astype(np.float32).copy()
    B, t = input_ids.shape
    # MinNewTokensLengthLogitsProcessor :250-260
    if spec.min_new_tokens > 0 and (t - prompt_len) < spec.min_new_tokens:
        s[:, spec.eos] = NEG_INF
    # SuppressTokensAtBeginLogitsProcessor :1859-1866
    if spec.begin_suppress and t == begin_index:
        s[:, spec.begin_suppress] = NEG_INF
    # SuppressTokensLogitsProcessor :1902-1906
    if spec.suppress:
        s[:, spec.suppress] = NEG_INF
    # WhisperTimeStampLogitsProcessor :2000-2047
    tb = spec.timestamp_begin
    s[:, spec.no_timestamps] = NEG_INF
    for k in range(B):
        seq = input_ids[k, begin_index:].tolist()
        last_was_ts = len(seq) >= 1 and seq[-1] >= tb
        penult_was_ts = len(seq) < 2 or seq[-2] >= tb
        if last_was_ts:
            if penult_was_ts:
                s[k, tb:] = NEG_INF
            else:
                s[k, :spec.eos] = NEG_INF
        ts = [x for x in seq if x >= tb]
        if ts:
            last = ts[-1] if (last_was_ts and not penult_was_ts) else ts[-1] + 1
            s[k, tb:last] = NEG_INF
    if t == begin_index:
        s[:, :tb] = NEG_INF
        if spec.max_initial_timestamp_index is not None:
            s[:, tb + spec.max_initial_timestamp_index + 1:] = NEG_INF
    return s


def process(spec: ProcessorSpec, input_ids: np.ndarray, scores: np.ndarray, begin_index: int,
            prompt_len: int) -> np.ndarray:
    """input_ids [B,t] (prompt + generated so far), scores [B,V] f32 -> processed [B,V]."""
    s = _masked(spec, input_ids, scores, begin_index, prompt_len)
    tb = spec.timestamp_begin
    lp = log_softmax(s)
    for k in range(s.shape[0]):
        if logsumexp(lp[k, tb:]) > lp[k, :tb].max():
            s[k, :tb] = NEG_INF
    return s


def timestamp_mass_margin(spec: ProcessorSpec, input_ids: np.ndarray, scores: np.ndarray, begin_index: int, prompt_len: int) -> np.ndarray:
    """[B] logsumexp(log p[timestamps]) - max(log p[text]) of the masked scores: the quantity the timestamp-mass rule
    (:2041-2045) compares with zero -- how far a row is from switching between "a timestamp is forced" and "text allowed"."""
    s = _masked(spec, input_ids, scores, begin_index, prompt_len)
    tb = spec.timestamp_begin
    lp = log_softmax(s)
    return np.asarray([logsumexp(lp[k, tb:]) - lp[k, :tb].max() for k in range(s.shape[0])], dtype=np.float32)
