"""Token -> word collation and chunk-seam merging (host side, seam 3 of SURVEY.md section 8b).

Same observable behaviour as ``WhisperTokenizer._decode_asr(..., return_timestamps="word")``
(TF/models/whisper/tokenization_whisper.py:901-1150) and its helpers (:1153-1406): stride-aware time
offsets, deferred timestamps inside strides, longest-common-token-subsequence seam merge constrained by
timestamp order, unicode/space word grouping, punctuation merging, 0.01 s rounding.  The algorithm runs natively
(``csrc/collate.cpp`` behind ``cw_collate_*`` of the C ABI); this module holds the ``Vocabulary`` table (id -> bytes,
no ``transformers`` object needed at run time) and the thin binding.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


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
        from .languages import LANGUAGES
        return cls(table, specials, tok.eos_token_id, tb, tok.convert_tokens_to_ids("<|startofprev|>"),
                   tok.convert_tokens_to_ids("<|startoftranscript|>"), dict(LANGUAGES), getattr(tok, "language", None))

    @classmethod
    def from_pretrained(cls, path: str) -> "Vocabulary":
        """Build the table from a checkpoint directory (or a ``tokenizer.json`` path) without ``transformers``:
        the byte-level BPE vocabulary of ``tokenizer.json`` (``model.vocab`` + ``added_tokens``; falls back to
        ``vocab.json`` + ``added_tokens.json``).  Decoding needs no merges: a token string maps to bytes through the
        GPT-2 unicode<->byte alphabet.  Replaces ``AutoProcessor.from_pretrained(...).tokenizer`` (REF/transcribe.py:19)."""
        import json
        import os
        from .languages import LANGUAGES
        from .synthetic import bytes_to_unicode
        base = path if os.path.isdir(path) else os.path.dirname(path)
        tj = path if os.path.isfile(path) else os.path.join(base, "tokenizer.json")
        ids: Dict[int, str] = {}
        special_ids = set()
        if os.path.exists(tj):
            j = json.load(open(tj, encoding="utf-8"))
            for s, i in j["model"]["vocab"].items():
                ids[int(i)] = s
            for a in j.get("added_tokens", []):
                ids[int(a["id"])] = a["content"]
                if a.get("special"):
                    special_ids.add(int(a["id"]))
        else:
            vj = os.path.join(base, "vocab.json")
            if not os.path.exists(vj):
                raise FileNotFoundError(f"neither tokenizer.json nor vocab.json under {base}")
            for s, i in json.load(open(vj, encoding="utf-8")).items():
                ids[int(i)] = s
            aj = os.path.join(base, "added_tokens.json")
            if os.path.exists(aj):
                for s, i in json.load(open(aj, encoding="utf-8")).items():
                    ids[int(i)] = s
        n = max(ids) + 1
        by_name = {s: i for i, s in ids.items()}
        if "<|notimestamps|>" not in by_name or "<|endoftext|>" not in by_name:
            raise ValueError("not a Whisper vocabulary: <|notimestamps|> / <|endoftext|> missing")
        tb = by_name["<|notimestamps|>"] + 1
        eos = by_name["<|endoftext|>"]
        u2b = {c: b for b, c in bytes_to_unicode().items()}
        table: List[Optional[bytes]] = []
        specials: Dict[int, str] = {}
        for i in range(n):
            s = ids.get(i)
            # specials: flagged in tokenizer.json, else every <|...|> tag below the timestamp range
            is_special = (i in special_ids) if special_ids else (s is not None and i < tb and s.startswith("<|") and s.endswith("|>"))
            if s is None or i >= tb:
                table.append(None)
            elif is_special:
                specials[i] = s
                table.append(None)
            else:
                try:
                    table.append(bytes(u2b[ch] for ch in s))
                except KeyError:
                    table.append(s.encode("utf-8"))
        return cls(table, specials, eos, tb, by_name.get("<|startofprev|>"), by_name.get("<|startoftranscript|>"),
                   dict(LANGUAGES), None)

    def text(self, ids: Sequence[int]) -> str:
        buf = bytearray()
        for i in ids:
            b = self.token_bytes[i]
            if b is not None:
                buf += b
        return buf.decode("utf-8", errors="replace")


# -------------------------------------------------------------------------------------------------
_NO_SPACE_LANGS = {"chinese", "japanese", "thai", "lao", "myanmar", "cantonese"}


def _native_vocab(vocab: "Vocabulary"):
    """Lazily builds the C-side table (cw_vocab_create) for a Vocabulary."""
    h = getattr(vocab, "_handle", None)
    if h:
        return h
    import ctypes as C
    from . import _native
    lib = _native.load()
    n = len(vocab.token_bytes)
    offsets = np.zeros(n + 1, dtype=np.int64)
    kind = np.full(n, 2, dtype=np.int8)
    lang_class = np.full(n, -1, dtype=np.int8)
    parts = []
    pos = 0
    for i, b in enumerate(vocab.token_bytes):
        if i in vocab.specials:
            kind[i] = 1
            lang = vocab.languages.get(vocab.specials[i][2:-2])
            if lang is not None:
                lang_class[i] = 1 if lang in _NO_SPACE_LANGS else 0
        elif b is not None:
            kind[i] = 0
            parts.append(b)
            pos += len(b)
        offsets[i + 1] = pos
    blob = np.frombuffer(b"".join(parts) or b"\x00", dtype=np.uint8).copy()
    default_class = 1 if (vocab.default_language in _NO_SPACE_LANGS) else 0
    h = lib.cw_vocab_create(n, blob.ctypes.data_as(C.c_void_p), offsets.ctypes.data_as(C.c_void_p),
                            kind.ctypes.data_as(C.c_void_p), lang_class.ctypes.data_as(C.c_void_p), int(vocab.eos),
                            int(vocab.timestamp_begin), -1 if vocab.startofprev is None else int(vocab.startofprev),
                            -1 if vocab.sot is None else int(vocab.sot), default_class)
    if not h:
        raise RuntimeError("cw_vocab_create failed")
    vocab._handle = h
    return h


def decode_asr(vocab: Vocabulary, model_outputs: List[dict], time_precision: float = 0.02, warn=None,
               return_timestamps="word"):
    """model_outputs: [{"tokens", "token_timestamps", optional "stride": (len_s, left_s, right_s)}] in audio
    order -> (text, [{"text", "timestamp": (start, end)}]).  Runs in libcrisperwhisper.so (csrc/collate.cpp).
    ``return_timestamps="word"``: word chunks; ``True``: one chunk per timestamp-delimited segment (token_timestamps
    not needed; a missing boundary is None, as in the reference)."""
    if return_timestamps not in ("word", True):
        raise ValueError("return_timestamps must be 'word' or True")
    seg = return_timestamps is True
    import ctypes as C
    from . import _native
    lib = _native.load()
    col = lib.cw_collate_begin(_native_vocab(vocab), float(time_precision))
    if not col:
        raise RuntimeError("cw_collate_begin failed")
    try:
        if seg:
            lib.cw_collate_set_mode(col, 1)
        for out in model_outputs:
            toks = np.ascontiguousarray(np.asarray(out["tokens"]).reshape(-1), dtype=np.int64)
            ts = np.ascontiguousarray(np.asarray(out.get("token_timestamps", np.zeros(0))).reshape(-1), dtype=np.float32)
            stride = out.get("stride")
            cl, sl, sr = (stride if stride is not None else (0.0, 0.0, 0.0))
            rc = lib.cw_collate_feed(col, toks.ctypes.data_as(C.c_void_p), len(toks), ts.ctypes.data_as(C.c_void_p),
                                     len(ts), 1 if stride is not None else 0, float(cl), float(sl), float(sr))
            if rc != 0:
                raise ValueError("word collation failed: token_timestamps shorter than tokens")
        nw, tb, wb, warned = C.c_int32(0), C.c_int64(0), C.c_int64(0), C.c_int32(0)
        lib.cw_collate_finish(col, C.byref(nw), C.byref(tb), C.byref(wb), C.byref(warned))
        if warned.value and warn is not None:
            warn("Whisper did not predict an ending timestamp, which can happen if audio is cut off in the "
                 "middle of a word. Also make sure WhisperTimeStampLogitsProcessor was used during generation.")
        n = nw.value
        text = np.zeros(max(tb.value, 1), dtype=np.uint8)
        blob = np.zeros(max(wb.value, 1), dtype=np.uint8)
        starts = np.zeros(max(n, 1), dtype=np.float64)
        ends = np.zeros(max(n, 1), dtype=np.float64)
        offs = np.zeros(n + 1, dtype=np.int64)
        lib.cw_collate_get(col, text.ctypes.data_as(C.c_void_p), starts.ctypes.data_as(C.c_void_p),
                           ends.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), blob.ctypes.data_as(C.c_void_p))
        raw = blob.tobytes()
        def _t(x):
            return None if x != x else float(x)                  # NaN = no timestamp predicted (segment mode)
        words = [{"text": raw[offs[k]:offs[k + 1]].decode("utf-8"), "timestamp": (_t(starts[k]), _t(ends[k]))}
                 for k in range(n)]
        return text.tobytes()[:tb.value].decode("utf-8"), words
    finally:
        lib.cw_collate_free(col)
