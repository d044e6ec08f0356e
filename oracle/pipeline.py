"""ASR pipeline driver, restatement (test oracle) of
TF/pipelines/automatic_speech_recognition.py:61-84 (chunk_iter), :432-448 (chunk/stride sizes),
:483-598 (_forward), :600-710 (postprocess) and TF/pipelines/pt_utils.py batching order.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from . import collate as C
from . import generate as G
from . import mel as M


def chunk_iter(n: int, chunk_len: int, stride_left: int, stride_right: int):
    """:61-84 -> list of (start, length, (chunk_len, left, right), is_last)."""
    out = []
    step = chunk_len - stride_left - stride_right
    for start in range(0, n, step):
        end = start + chunk_len
        length = min(end, n) - start
        sl = 0 if start == 0 else stride_left
        is_last = end >= n
        sr = 0 if is_last else stride_right
        if length > sl:
            out.append((start, length, (length, sl, sr), is_last))
        if is_last:
            break
    return out


def transcribe(model, spec: G.GenSpec, vocab: C.ByteVocab, pcm: np.ndarray, *, n_mels: int, batch_size: int = 16,
               chunk_length_s: float = 30.0, language: Optional[str] = "<|en|>", task: Optional[str] = "transcribe",
               max_new_tokens: Optional[int] = None, min_new_tokens: Optional[int] = None,
               sampling_rate: int = 16000, trace: Optional[list] = None):
    """The ``pipe(array)`` call of REF/transcribe.py:33 on a mono float array -> {"text","chunks"}."""
    pcm = np.asarray(pcm, dtype=np.float32)
    chunk_len = int(round(chunk_length_s * sampling_rate))
    stride = int(round(chunk_length_s / 6 * sampling_rate))
    items = chunk_iter(len(pcm), chunk_len, stride, stride)
    outputs: List[dict] = []
    for b0 in range(0, len(items), batch_size):
        batch = items[b0:b0 + batch_size]
        pcm_b = np.zeros((len(batch), M.N_SAMPLES), dtype=np.float32)
        nf = np.zeros(len(batch), dtype=np.int64)
        for k, (start, length, _, _) in enumerate(batch):
            pcm_b[k], nv = M.pad_or_trim(pcm[start:start + length])
            nf[k] = M.attention_mask_frames(nv)
        feats = M.log_mel(pcm_b, n_mels)
        out = G.generate(model, spec, feats, nf, language=language, task=task,
                         max_new_tokens=max_new_tokens, min_new_tokens=min_new_tokens, trace=trace)
        for k, (_, _, st, _) in enumerate(batch):
            outputs.append({"tokens": out["sequences"][k], "token_timestamps": out["token_timestamps"][k],
                            "stride": tuple(x / sampling_rate for x in st)})
    text, words = C.decode_asr(vocab, outputs, time_precision=0.02)
    return {"text": text, "chunks": words}
