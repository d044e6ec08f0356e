"""Audio ingest + chunk scheduling (host, integer index math).

``chunk_windows`` follows TF/pipelines/automatic_speech_recognition.py:61-84 (chunk_iter) and
:432-448 (chunk/stride lengths): 30 s windows, chunk_length_s/6 stride each side, hop = chunk - 2*stride.
``read_audio`` replaces the ffmpeg subprocess of TF/pipelines/audio_utils.py:9-45 for WAV input
(ffmpeg is not required; other containers raise like the reference does when ffmpeg is missing).
"""
from __future__ import annotations

import io
from typing import List, Tuple

import numpy as np

SAMPLING_RATE = 16000


def chunk_windows(n_samples: int, chunk_len: int, stride_left: int, stride_right: int):
    """-> [(start, length, (length, left, right), is_last)] in audio order."""
    if chunk_len < stride_left + stride_right:
        raise ValueError("Chunk length must be superior to stride length")
    step = chunk_len - stride_left - stride_right
    out: List[Tuple[int, int, Tuple[int, int, int], bool]] = []
    for start in range(0, n_samples, step):
        end = start + chunk_len
        length = min(end, n_samples) - start
        left = 0 if start == 0 else stride_left
        last = end >= n_samples
        right = 0 if last else stride_right
        if length > left:
            out.append((start, length, (length, left, right), last))
        if last:
            break
    return out


def resample(x: np.ndarray, sr_in: int, sr_out: int = SAMPLING_RATE) -> np.ndarray:
    if sr_in == sr_out:
        return x
    from math import gcd
    from scipy.signal import resample_poly
    g = gcd(int(sr_in), int(sr_out))
    return resample_poly(x.astype(np.float64), sr_out // g, sr_in // g).astype(np.float32)


def decode_wav_bytes(data: bytes, sampling_rate: int = SAMPLING_RATE) -> np.ndarray:
    from scipy.io import wavfile
    try:
        sr, x = wavfile.read(io.BytesIO(data))
    except Exception as e:
        raise ValueError("Soundfile is either not in the correct format or is malformed. Only RIFF/WAV input is "
                         "decoded natively (the reference needs ffmpeg for anything else).") from e
    if x.dtype.kind == "i":
        x = x.astype(np.float32) / float(np.iinfo(x.dtype).max + 1)
    elif x.dtype.kind == "u":
        x = (x.astype(np.float32) - 128.0) / 128.0
    else:
        x = x.astype(np.float32)
    if x.ndim == 2:
        x = x.mean(axis=1)                      # mono mixdown
    return resample(x, sr, sampling_rate)


def read_audio(path: str, sampling_rate: int = SAMPLING_RATE) -> np.ndarray:
    with open(path, "rb") as f:
        return decode_wav_bytes(f.read(), sampling_rate)
