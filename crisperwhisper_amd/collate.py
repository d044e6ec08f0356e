"""Token -> word collation and chunk-seam merging (host side, seam 3 of SURVEY.md section 8b).

Same observable behaviour as ``WhisperTokenizer._decode_asr(..., return_timestamps="word")``
(TF/models/whisper/tokenization_whisper.py:901-1150) and its helpers (:1153-1406): stride-aware time
offsets, deferred timestamps inside strides, longest-common-token-subsequence seam merge constrained by
timestamp order, unicode/space word grouping, punctuation merging, 0.01 s rounding.  Works on a plain
``Vocabulary`` (id -> bytes) so it needs no ``transformers`` object at run time.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

_NO_SPACE_LANGS = {"chinese", "japanese", "thai", "lao", "myanmar", "cantonese"}
_PUNCT = "!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~"
_PREPEND = "\"'“¡¿([{-"
_APPEND = "\"'.。,，!！?？:：”)]}、"
_REPL = "�"


class Vocabulary:
    """id -> raw bytes for text tokens; specials carry their ``<|name|>`` string."""

    def __init__(self, token_bytes: Sequence[Optional[bytes]], specials: Dict[int, str], eos: int,
                 timestamp_begin: int, startofprev: Optional[int], sot: Optional[int],
                 languages: Optional[Dict[str, str]] = None, default_language: Optional[str] = None):
        self.token_bytes = list(token_bytes)
        self.specials = dict(specials)
        self.eos = eos
        self.timestamp_begin = timestamp_begin
        self.startofprev = startofprev
        self.sot = sot
        self.languages = languages or {}
        self.default_language = default_language

    @classmethod
    def from_synthetic(cls, v) -> "Vocabulary":
        names = v.special_names()
        return cls(v.token_bytes(), {v.eos + i: n for i, n in enumerate(names)}, v.eos, v.timestamp_begin,
                   v.startofprev, v.sot, {"en": "english", "zh": "chinese", "de": "german", "es": "spanish"})

    @classmethod
    def from_hf_tokenizer(cls, tok) -> "Vocabulary":
        """Build the table from a ``WhisperTokenizer`` (byte-level BPE: token string -> bytes through the
        GPT-2 unicode<->byte alphabet)."""
        from .synthetic import bytes_to_unicode
        u2b = {c: b for b, c in bytes_to_unicode().items()}
        special_ids = set(tok.all_special_ids)
        tb = tok.convert_tokens_to_ids("<|notimestamps|>") + 1
        n = len(tok)
        strings = tok.convert_ids_to_tokens(list(range(n)))
        table: List[Optional[bytes]] = []
        specials: Dict[int, str] = {}
        for i, s in enumerate(strings):
            if i in special_ids:
                specials[i] = s
                table.append(None)
            elif i >= tb or s is None:
                table.append(None)
            else:
                try:
                    table.append(bytes(u2b[ch] for ch in s))
                except KeyError:
                    table.append(s.encode("utf-8"))
        try:
            from transformers.models.whisper.tokenization_whisper import LANGUAGES
            langs = dict(LANGUAGES)
        except Exception:  # pragma: no cover
            langs = {}
        return cls(table, specials, tok.eos_token_id, tb, tok.convert_tokens_to_ids("<|startofprev|>"),
                   tok.convert_tokens_to_ids("<|startoftranscript|>"), langs, getattr(tok, "language", None))

    def text(self, ids: Sequence[int]) -> str:
        buf = bytearray()
        for i in ids:
            b = self.token_bytes[i]
            if b is not None:
                buf += b
        return buf.decode("utf-8", errors="replace")


# -------------------------------------------------------------------------------------------------
def merge_overlapping(seqs: List[List[int]], ts_seqs: List[List[Tuple[float, float]]]):
    """Greedy pairwise seam merge: for every overlap length pick the alignment with the best
    (matches / length + length * 1e-4) score among those with > 1 match whose left timestamps do not
    exceed the right ones; cut both sides at the middle of the matched window."""
    left, left_ts = seqs[0], ts_seqs[0]
    merged: List[int] = []
    merged_ts: List[Tuple[float, float]] = []
    for right, right_ts in zip(seqs[1:], ts_seqs[1:]):
        nl, nr = len(left), len(right)
        best, window = 0.0, (nl, nl, 0, 0)
        for span in range(1, nl + nr):
            l0, l1 = max(0, nl - span), min(nl, nl + nr - span)
            r0, r1 = max(0, span - nl), min(nr, span)
            if l1 - l0 != r1 - r0:
                raise RuntimeError("There is a bug within whisper `decode_asr` function, please report it. "
                                   "Dropping to prevent bad inference.")
            hits = 0
            for k in range(l1 - l0):
                if left[l0 + k] == right[r0 + k] and left_ts[l0 + k] <= right_ts[r0 + k]:
                    hits += 1
            score = hits / span + span / 10000.0
            if hits > 1 and score > best:
                best, window = score, (l0, l1, r0, r1)
        l0, l1, r0, r1 = window
        lmid, rmid = (l1 + l0) // 2, (r1 + r0) // 2
        merged.extend(left[:lmid])
        merged_ts.extend(left_ts[:lmid])
        left, left_ts = right[rmid:], right_ts[rmid:]
    merged.extend(left)
    merged_ts.extend(left_ts)
    return merged, merged_ts


def _unicode_pieces(vocab: Vocabulary, tokens: List[int]):
    whole = vocab.text(tokens)
    pieces, piece_idx = [], []
    cur, cur_idx, consumed = [], [], 0
    for k, t in enumerate(tokens):
        cur.append(t)
        cur_idx.append(k)
        s = vocab.text(cur)
        p = s.find(_REPL)
        if p < 0 or consumed + p >= len(whole) or whole[consumed + p] == _REPL:
            pieces.append(s)
            piece_idx.append(cur_idx)
            consumed += len(s)
            cur, cur_idx = [], []
    return pieces, piece_idx


def words_from_tokens(vocab: Vocabulary, tokens: List[int], language: Optional[str]):
    """Group tokens into words; returns (words, token index lists)."""
    pieces, piece_idx = _unicode_pieces(vocab, tokens)
    if language in _NO_SPACE_LANGS:
        words, idx = pieces, piece_idx
    else:
        words, idx = [], []
        for s, ix in zip(pieces, piece_idx):
            starts_word = (tokens[ix[0]] >= vocab.eos) or s.startswith(" ") or (s.strip() in _PUNCT) or not words
            if starts_word:
                words.append(s)
                idx.append(list(ix))
            else:
                words[-1] += s
                idx[-1].extend(ix)
    # punctuation that belongs to the following word ...
    j = len(words) - 1
    for i in range(len(words) - 2, -1, -1):
        if words[i].startswith(" ") and words[i].strip() in _PREPEND:
            words[j] = words[i] + words[j]
            idx[j] = idx[i] + idx[j]
            words[i], idx[i] = "", []
        else:
            j = i
    # ... and to the preceding one
    i = 0
    for j in range(1, len(words)):
        if not words[i].endswith(" ") and words[j] in _APPEND:
            words[i] += words[j]
            idx[i] = idx[i] + idx[j]
            words[j], idx[j] = "", []
        else:
            i = j
    keep = [k for k, w in enumerate(words) if w]
    return [words[k] for k in keep], [idx[k] for k in keep]


class WordCollator:
    """Streaming state machine over chunk outputs (in audio order)."""

    def __init__(self, vocab: Vocabulary, time_precision: float = 0.02, segment_size: int = 1500):
        self.v = vocab
        self.tp = time_precision
        self.segment_size = segment_size
        self.words: List[dict] = []
        self.text_parts: List[str] = []
        self.language: Optional[str] = None
        self.time_offset = 0.0
        self.pending: List[List[int]] = []
        self.pending_ts: List[List[Tuple[float, float]]] = []
        self.open_start: Optional[float] = None
        self.skip = False

    def _flush(self):
        toks, ts = merge_overlapping(self.pending, self.pending_ts)
        self.text_parts.append(self.v.text(toks))
        lang = self.language or self.v.default_language or "english"
        words, idx = words_from_tokens(self.v, toks, lang)
        for w, ix in zip(words, idx):
            self.words.append({"text": w, "timestamp": (ts[ix[0]][0], ts[ix[-1]][1])})
        self.pending, self.pending_ts = [], []
        self.open_start = None

    def feed(self, tokens: Sequence[int], token_timestamps: Sequence[float],
             stride: Optional[Tuple[float, float, float]] = None):
        v, tp = self.v, self.tp
        tb = v.timestamp_begin
        ids = [int(t) for t in tokens]
        if ids and v.startofprev is not None and ids[0] == v.startofprev:        # _strip_prompt
            ids = ids[ids.index(v.sot):] if v.sot in ids else []
        tts = [float(t) for t in token_timestamps]
        deferred_from = None          # first timestamp token inside the right stride
        first_ts = tb
        chunk_len = stride_right = None
        if stride is not None:
            chunk_len, stride_left, stride_right = stride
            self.time_offset -= stride_left
            right_start = chunk_len - stride_right
            if stride_left:
                first_ts = stride_left / tp + tb
            if stride_right:
                for t in reversed(ids):
                    if t >= tb:
                        if deferred_from is not None and (t - tb) * tp < right_start:
                            break
                        deferred_from = t
        cur, cur_ts = [], []
        cur_max = prev_len = penult = 0.0
        for i, t in enumerate(ids):
            if t in v.specials:
                lang = v.languages.get(v.specials[t][2:-2])
                if lang is not None:
                    self.language = lang
            elif t >= tb:
                stamp = float((t - tb) * tp)
                if stamp < cur_max:                                   # a new 30 s window inside this output
                    single_end = i >= 2 and not (ids[i - 1] >= tb and ids[i - 2] >= tb)
                    if single_end:
                        prev_len += tp * self.segment_size
                    else:
                        cur_max = penult
                        prev_len += penult
                penult = cur_max
                cur_max = stamp
                when = round((t - tb) * tp + self.time_offset + prev_len, 2)
                if deferred_from and t >= deferred_from:
                    self.skip = True
                elif self.skip or (self.pending and t < first_ts):
                    self.skip = False
                elif self.open_start is None:
                    self.open_start = when
                elif when != self.open_start:
                    self.pending.append(cur)
                    self.pending_ts.append(cur_ts)
                    self._flush()
                    cur, cur_ts = [], []
            else:
                cur.append(t)
                begin = round(0.0 + self.time_offset, 2) if i == 0 else round(tts[i - 1] + self.time_offset, 2)
                cur_ts.append((begin, round(tts[i] + self.time_offset, 2)))
        if stride is not None:
            self.time_offset += chunk_len - stride_right
        if cur:
            self.pending.append(cur)
            self.pending_ts.append(cur_ts)
        elif not any(p for p in self.pending):
            self.pending, self.pending_ts = [], []
            self.open_start = None

    def finish(self, warn=None):
        if self.pending:
            if warn is not None:
                warn("Whisper did not predict an ending timestamp, which can happen if audio is cut off in the "
                     "middle of a word. Also make sure WhisperTimeStampLogitsProcessor was used during generation.")
            self._flush()
        return "".join(self.text_parts), self.words


def decode_asr(vocab: Vocabulary, model_outputs: List[dict], time_precision: float = 0.02, warn=None):
    """model_outputs: [{"tokens", "token_timestamps", optional "stride": (len_s, left_s, right_s)}]."""
    wc = WordCollator(vocab, time_precision)
    for out in model_outputs:
        wc.feed(out["tokens"], out["token_timestamps"], out.get("stride"))
    return wc.finish(warn)
