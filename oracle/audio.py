"""CPU restatement of the audio ingest in front of the feature extractor -- TEST INFRASTRUCTURE ONLY.

Follows:
  * ``torchaudio.functional.resample`` with its defaults (``sinc_interp_hann``, ``lowpass_filter_width=6``,
    ``rolloff=0.99``), the resampler the reference reaches through
    TF/pipelines/automatic_speech_recognition.py:398-412 (``F.resample``) and REF/app.py:94-95 (``T.Resample``).
    torchaudio is a third-party dependency of the reference (REF/requirements.txt, unpinned) and is NOT installed in
    this image, so this file restates its published algorithm (``_get_sinc_resample_kernel`` /
    ``_apply_sinc_resample_kernel``, torchaudio 2.x functional.py) -- **parity unpinned**: no torchaudio output is
    available offline to pin it; the tests anchor it on analytic properties (identity, length formula, in-band sinusoids
    keep amplitude and phase, out-of-band ones are rejected) and, round 3, on the published closed-form kernel evaluated
    sample by sample in f64 by a route that shares no code with this file
    (tests/test_audio_ingest.py::test_resampler_equals_the_closed_form_kernel_by_direct_convolution).
  * the sample decoding of ``ffmpeg -ac 1 -f f32le`` (TF/pipelines/audio_utils.py:9-45): integer PCM scaled to
    [-1, 1), channels averaged.
  * REF/app.py:85-93: ``(y - mean) / std / 8``.
"""
from __future__ import annotations

import math

import numpy as np

PCM_U8, PCM_S16, PCM_S24, PCM_S32, PCM_F32, PCM_F64 = range(6)
LOWPASS_FILTER_WIDTH = 6
ROLLOFF = 0.99


def resample_taps(sr_in: int, sr_out: int):
    """-> (K [new, 2*width + orig] float32, orig, new, width); f64 arithmetic rounded once (dtype=None branch)."""
    g = math.gcd(int(sr_in), int(sr_out))
    orig, new = int(sr_in) // g, int(sr_out) // g
    base_freq = min(orig, new) * ROLLOFF
    width = math.ceil(LOWPASS_FILTER_WIDTH * orig / base_freq)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t *= base_freq
    t = np.clip(t, -LOWPASS_FILTER_WIDTH, LOWPASS_FILTER_WIDTH)
    window = np.cos(t * math.pi / LOWPASS_FILTER_WIDTH / 2) ** 2
    t *= math.pi
    scale = base_freq / orig
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(t == 0, 1.0, np.sin(t) / t)
    k = k * window * scale
    return k.astype(np.float32), orig, new, width


def resampled_length(n: int, sr_in: int, sr_out: int) -> int:
    g = math.gcd(int(sr_in), int(sr_out))
    return -(-(n * (int(sr_out) // g)) // (int(sr_in) // g))


def resample(x: np.ndarray, sr_in: int, sr_out: int) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    if int(sr_in) == int(sr_out):
        return x
    k, orig, new, width = resample_taps(sr_in, sr_out)
    n = len(x)
    xp = np.concatenate([np.zeros(width, np.float32), x, np.zeros(width + orig, np.float32)])
    n_frames = (len(xp) - k.shape[1]) // orig + 1                      # conv1d, stride = orig
    win = np.lib.stride_tricks.as_strided(xp, shape=(n_frames, k.shape[1]), strides=(xp.strides[0] * orig, xp.strides[0]))
    out = (win.astype(np.float64) @ k.astype(np.float64).T).astype(np.float32)   # [frames, new]; f64 accumulate: the exact value of the f32 taps
    return out.reshape(-1)[: resampled_length(n, sr_in, sr_out)]


def pcm_to_mono(raw: bytes, fmt: int, channels: int) -> np.ndarray:
    b = np.frombuffer(raw, dtype=np.uint8)
    if fmt == PCM_U8:
        x = (b.astype(np.float32) - 128.0) / 128.0
    elif fmt == PCM_S16:
        x = b.view("<i2").astype(np.float32) / 32768.0
    elif fmt == PCM_S24:
        t = b.reshape(-1, 3).astype(np.int32)
        v = t[:, 0] | (t[:, 1] << 8) | (t[:, 2] << 16)
        v = np.where(v & 0x800000, v - (1 << 24), v)
        x = v.astype(np.float32) / 8388608.0
    elif fmt == PCM_S32:
        x = b.view("<i4").astype(np.float32) / 2147483648.0
    elif fmt == PCM_F32:
        x = b.view("<f4").astype(np.float32)
    elif fmt == PCM_F64:
        x = b.view("<f8").astype(np.float32)
    else:
        raise ValueError(fmt)
    if channels > 1:
        x = x.reshape(-1, channels)
        s = np.zeros(len(x), np.float32)
        for c in range(channels):
            s = s + x[:, c]
        x = s / np.float32(channels)
    return x


def normalise(y: np.ndarray) -> np.ndarray:
    y = np.asarray(y, dtype=np.float32)
    return ((y - np.mean(y)) / np.std(y)) / 8
