"""Log-mel front end, numpy restatement (test oracle; see oracle/__init__.py).

Follows TF/models/whisper/feature_extraction_whisper.py:135-168 (``_torch_extract_fbank_features``),
:296-341 (padding + attention mask) and TF/audio_utils.py:448-560, 638-729 (slaney mel bank).
"""
from __future__ import annotations

import numpy as np

N_FFT = 400
HOP = 160
N_SAMPLES = 480000
N_FRAMES = 3000
SAMPLING_RATE = 16000


def hertz_to_mel_slaney(freq):
    """TF/audio_utils.py:470-481."""
    freq = np.asarray(freq, dtype=np.float64)
    mels = 3.0 * freq / 200.0
    logstep = 27.0 / np.log(6.4)
    log_region = freq >= 1000.0
    safe = np.where(log_region, freq, 1000.0)
    return np.where(log_region, 15.0 + np.log(safe / 1000.0) * logstep, mels)


def mel_to_hertz_slaney(mels):
    """TF/audio_utils.py:505-517."""
    mels = np.asarray(mels, dtype=np.float64)
    freq = 200.0 * mels / 3.0
    logstep = np.log(6.4) / 27.0
    log_region = mels >= 15.0
    return np.where(log_region, 1000.0 * np.exp(logstep * (mels - 15.0)), freq)


def mel_filter_bank(n_mels: int, n_bins: int = 201, sr: int = SAMPLING_RATE,
                    fmin: float = 0.0, fmax: float = 8000.0) -> np.ndarray:
    """[n_bins, n_mels] float64, slaney scale + slaney norm (TF/audio_utils.py:700-720)."""
    mel_freqs = np.linspace(hertz_to_mel_slaney(fmin), hertz_to_mel_slaney(fmax), n_mels + 2)
    filter_freqs = mel_to_hertz_slaney(mel_freqs)
    fft_freqs = np.linspace(0, sr // 2, n_bins)
    filter_diff = np.diff(filter_freqs)
    slopes = filter_freqs[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / filter_diff[:-1]
    up = slopes[:, 2:] / filter_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (filter_freqs[2:n_mels + 2] - filter_freqs[:n_mels])
    return fb * enorm[None, :]


def hann_periodic(n: int = N_FFT) -> np.ndarray:
    """torch.hann_window(n) (periodic), :141."""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n))


def pad_or_trim(pcm: np.ndarray, n_samples: int = N_SAMPLES):
    """:296-307: right-pad with zeros / truncate to 30 s; returns (padded, n_valid)."""
    pcm = np.asarray(pcm, dtype=np.float32)
    n_valid = min(len(pcm), n_samples)
    out = np.zeros(n_samples, dtype=np.float32)
    out[:n_valid] = pcm[:n_valid]
    return out, n_valid


def attention_mask_frames(n_valid: int) -> int:
    """:332-341: mask[:, ::hop] -> number of valid mel frames."""
    return (n_valid + HOP - 1) // HOP


def log_mel(pcm_batch: np.ndarray, n_mels: int = 128, dtype=np.float64) -> np.ndarray:
    """[B, 480000] f32 -> [B, n_mels, 3000] f32.

    ``dtype`` selects the STFT precision (float64 = mathematically cleanest; float32 mimics
    torch.stft's arithmetic class).  Everything after the power spectrum is float32 like TF.
    """
    pcm_batch = np.asarray(pcm_batch, dtype=np.float32)
    if pcm_batch.ndim == 1:
        pcm_batch = pcm_batch[None]
    win = hann_periodic().astype(dtype)
    fb = mel_filter_bank(n_mels).astype(np.float32)  # TF stores float64, casts to f32 at :156
    out = np.empty((pcm_batch.shape[0], n_mels, N_FRAMES), dtype=np.float32)
    for b, x in enumerate(pcm_batch):
        xp = np.pad(x.astype(dtype), (N_FFT // 2, N_FFT // 2), mode="reflect")  # center=True
        idx = np.arange(N_FFT)[None, :] + HOP * np.arange(N_FRAMES)[:, None]     # drop frame 3000 (:154)
        frames = xp[idx] * win[None, :]
        spec = np.fft.rfft(frames, axis=1)
        power = (spec.real.astype(np.float64) ** 2 + spec.imag.astype(np.float64) ** 2)
        power = power.astype(np.float32)                                          # [3000, 201]
        mel = fb.T @ power.T                                                      # [n_mels, 3000] f32
        log_spec = np.log10(np.maximum(mel, np.float32(1e-10))).astype(np.float32)
        log_spec = np.maximum(log_spec, log_spec.max() - np.float32(8.0))
        out[b] = (log_spec + np.float32(4.0)) / np.float32(4.0)
    return out
