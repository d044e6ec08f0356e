"""Token -> word collation with chunk-seam merging, restatement (test oracle) of
TF/models/whisper/tokenization_whisper.py:901-1150 (_decode_asr), :1153-1270
(_find_longest_common_sequence), :1273-1406 (word grouping helpers).

The tokenizer is abstracted as a ``ByteVocab``: id -> raw bytes for text tokens (byte-level BPE
decodes by concatenating token bytes and utf-8 decoding with replacement), plus the special ids.
"""
from __future__ import annotations

from typing import List, Optional

LANGUAGES = {"en": "english", "zh": "chinese", "de": "german", "es": "spanish"}
PUNCT = "!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~"
PREPEND = "\"'“¡¿([{-"
APPEND = "\"'.。,，!！?？:：”)]}、"


class ByteVocab:
    def __init__(self, token_bytes: List[Optional[bytes]], special_names: List[str], eos: int,
                 timestamp_begin: int, startofprev: int, sot: int):
        self.token_bytes = token_bytes
        self.eos = eos
        self.timestamp_begin = timestamp_begin
        self.startofprev = startofprev
        self.sot = sot
        self.special = {eos + i: n for i, n in enumerate(special_names)}

    def decode(self, ids) -> str:
        return b"".join(self.token_bytes[i] for i in ids if self.token_bytes[i] is not None).decode("utf-8", errors="replace")


def find_longest_common_sequence(sequences, ts_sequences=None):
    """:1153-1270."""
    left = sequences[0]
    left_len = len(left)
    total = []
    if ts_sequences:
        left_ts = ts_sequences[0]
        total_ts = []
    for si, right in enumerate(sequences[1:]):
        max_ = 0.0
        max_idx = (left_len, left_len, 0, 0)
        right_len = len(right)
        for i in range(1, left_len + right_len):
            eps = i / 10000.0
            ls, le = max(0, left_len - i), min(left_len, left_len + right_len - i)
            rs, re_ = max(0, i - left_len), min(right_len, i)
            l, r = left[ls:le], right[rs:re_]
            if len(l) != len(r):
                raise RuntimeError("bug in decode_asr")
            if ts_sequences:
                matches = sum(1 for k, e in enumerate(l)
                              if e == r[k] and left_ts[ls + k] <= ts_sequences[si + 1][rs + k])
            else:
                matches = sum(1 for a, b in zip(l, r) if a == b)
            matching = matches / i + eps
            if matches > 1 and matching > max_:
                max_ = matching
                max_idx = (ls, le, rs, re_)
        ls, le, rs, re_ = max_idx
        lmid, rmid = (le + ls) // 2, (re_ + rs) // 2
        total.extend(left[:lmid])
        left = right[rmid:]
        left_len = len(left)
        if ts_sequences:
            total_ts.extend(left_ts[:lmid])
            left_ts = ts_sequences[si + 1][rmid:]
    total.extend(left)
    if ts_sequences is None:
        return total
    if len(ts_sequences) > 0:
        total_ts.extend(left_ts)
        return total, total_ts
    return total, []


def split_tokens_on_unicode(tok: ByteVocab, tokens):
    """:1315-1344."""
    full = tok.decode(tokens)
    rc = "�"
    words, wtok, widx = [], [], []
    cur, curi = [], []
    off = 0
    for ti, t in enumerate(tokens):
        cur.append(t); curi.append(ti)
        dec = tok.decode(cur)
        if rc not in dec or off + dec.index(rc) >= len(full) or full[off + dec.index(rc)] == rc:
            words.append(dec); wtok.append(cur); widx.append(curi)
            cur, curi = [], []
            off += len(dec)
    return words, wtok, widx


def split_tokens_on_spaces(tok: ByteVocab, tokens):
    """:1347-1368."""
    sub, subt, subi = split_tokens_on_unicode(tok, tokens)
    words, wtok, widx = [], [], []
    for s, st, si in zip(sub, subt, subi):
        special = st[0] >= tok.eos
        with_space = s.startswith(" ")
        punct = s.strip() in PUNCT
        if special or with_space or punct or len(words) == 0:
            words.append(s); wtok.append(st); widx.append(si)
        else:
            words[-1] = words[-1] + s; wtok[-1].extend(st); widx[-1].extend(si)
    return words, wtok, widx


def merge_punctuations(words, tokens, indices, prepended=PREPEND, appended=APPEND):
    """:1371-1406."""
    i, j = len(words) - 2, len(words) - 1
    while i >= 0:
        if words[i].startswith(" ") and words[i].strip() in prepended:
            words[j] = words[i] + words[j]; tokens[j] = tokens[i] + tokens[j]; indices[j] = indices[i] + indices[j]
            words[i] = ""; tokens[i] = []; indices[i] = []
        else:
            j = i
        i -= 1
    i, j = 0, 1
    while j < len(words):
        if not words[i].endswith(" ") and words[j] in appended:
            words[i] += words[j]; tokens[i] += tokens[j]; indices[i] += indices[j]
            words[j] = ""; tokens[j] = []; indices[j] = []
        else:
            i = j
        j += 1
    words[:] = [w for w in words if w]
    tokens[:] = [t for t in tokens if t]
    indices[:] = [x for x in indices if x]


def collate_word_timestamps(tok, tokens, token_ts, language):
    """:1273-1312 (space-splitting languages only: the synthetic vocab is 'english')."""
    if language in {"chinese", "japanese", "thai", "lao", "myanmar", "cantonese"}:
        words, wt, wi = split_tokens_on_unicode(tok, tokens)
    else:
        words, wt, wi = split_tokens_on_spaces(tok, tokens)
    merge_punctuations(words, wt, wi)
    return [{"text": w, "timestamp": (token_ts[ix[0]][0], token_ts[ix[-1]][1])} for w, ix in zip(words, wi)]


def strip_prompt(ids, prompt_id, sot):
    """:725-743."""
    if not isinstance(ids, list):
        ids = list(ids)
    if ids and ids[0] == prompt_id:
        if sot in ids:
            return ids[ids.index(sot):]
        return []
    return ids


def decode_asr(tok: ByteVocab, model_outputs, time_precision=0.02, segment_size=1500, return_timestamps="word"):
    """:901-1150 with return_timestamps="word" (default) or True (segment chunks), return_language=None.

    model_outputs: list of {"tokens": [L] ints, "token_timestamps": [L'] floats (word mode only),
    optional "stride": (chunk_len_s, left_s, right_s)}.  Returns (text, chunks): word chunks, or for
    return_timestamps=True one {"text", "timestamp": (start, end)} per timestamp-delimited segment
    (end is None when Whisper did not predict an ending timestamp)."""
    word = return_timestamps == "word"
    last_language = None

    def new_chunk():
        return {"language": last_language, "timestamp": [None, None], "text": ""}

    chunks = []
    chunk = new_chunk()
    time_offset = 0.0
    tb = tok.timestamp_begin
    previous_tokens, previous_ts = [], []
    skip = False
    right_stride_start = None
    for output in model_outputs:
        token_ids = strip_prompt([int(t) for t in output["tokens"]], tok.startofprev, tok.sot)
        token_timestamps = [float(t) for t in output["token_timestamps"]] if word else []
        last_timestamp = None
        first_timestamp = tb
        cur_max_timestamp = 0.0
        prev_segments_len = 0.0
        penultimate_timestamp = 0.0
        if "stride" in output:
            chunk_len, stride_left, stride_right = output["stride"]
            time_offset -= stride_left
            right_stride_start = chunk_len - stride_right
            if stride_left:
                first_timestamp = stride_left / time_precision + tb
            if stride_right:
                for token in reversed(token_ids):
                    if token >= tb:
                        if last_timestamp is not None and (token - tb) * time_precision < right_stride_start:
                            break
                        last_timestamp = token
        current_tokens, current_ts = [], []
        for i, token in enumerate(token_ids):
            if token in tok.special:
                text = tok.special[token][2:-2]
                language = LANGUAGES.get(text)
                if language is not None:
                    chunk["language"] = language
                    last_language = language
            elif token >= tb:
                timestamp = float((token - tb) * time_precision)
                if timestamp < cur_max_timestamp:
                    last_was_single_ending = i >= 2 and not (token_ids[i - 1] >= tb and token_ids[i - 2] >= tb)
                    if last_was_single_ending:
                        prev_segments_len += time_precision * segment_size
                    else:
                        cur_max_timestamp = penultimate_timestamp
                        prev_segments_len += penultimate_timestamp
                penultimate_timestamp = cur_max_timestamp
                cur_max_timestamp = timestamp
                time = (token - tb) * time_precision + time_offset + prev_segments_len
                time = round(time, 2)
                if last_timestamp and token >= last_timestamp:
                    skip = True
                elif skip or (previous_tokens and token < first_timestamp):
                    skip = False
                elif chunk["timestamp"][0] is None:
                    chunk["timestamp"][0] = time
                else:
                    if time == chunk["timestamp"][0]:
                        pass
                    else:
                        chunk["timestamp"][1] = time
                        previous_tokens.append(current_tokens)
                        if word:
                            previous_ts.append(current_ts)
                        rt, rts = find_longest_common_sequence(previous_tokens, previous_ts)
                        chunk["text"] = tok.decode(rt)
                        if word:
                            chunk["words"] = collate_word_timestamps(tok, rt, rts, last_language or "english")
                        chunks.append(chunk)
                        previous_tokens, current_tokens, previous_ts, current_ts = [], [], [], []
                        chunk = new_chunk()
            else:
                current_tokens.append(token)
                if word:
                    if i == 0:
                        start_time = round(0.0 + time_offset, 2)
                    else:
                        start_time = round(token_timestamps[i - 1] + time_offset, 2)
                    end_time = round(token_timestamps[i] + time_offset, 2)
                    current_ts.append((start_time, end_time))
        if "stride" in output:
            time_offset += chunk_len - stride_right
        if current_tokens:
            previous_tokens.append(current_tokens)
            if word:
                previous_ts.append(current_ts)
        elif not any(p for p in previous_tokens):
            chunk = new_chunk()
            previous_tokens, current_tokens, previous_ts, current_ts = [], [], [], []
    if previous_tokens:
        rt, rts = find_longest_common_sequence(previous_tokens, previous_ts)
        chunk["text"] = tok.decode(rt)
        if word:
            chunk["words"] = collate_word_timestamps(tok, rt, rts, last_language or "english")
        chunks.append(chunk)
    full_text = "".join(c["text"] for c in chunks)
    if not word:
        return full_text, [{"text": c["text"], "timestamp": tuple(c["timestamp"])} for c in chunks]
    words = []
    for c in chunks:
        words.extend(c["words"])
    return full_text, words
