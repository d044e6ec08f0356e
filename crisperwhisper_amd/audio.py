"""Audio ingest + chunk scheduling (host, integer index math).

``chunk_windows`` follows TF/pipelines/automatic_speech_recognition.py:61-84 (chunk_iter) and
:432-448 (chunk/stride lengths): 30 s windows, chunk_length_s/6 stride each side, hop = chunk - 2*stride.
``read_audio`` / ``decode_wav_bytes`` replace the ffmpeg subprocess of TF/pipelines/audio_utils.py:9-45 for WAV and FLAC
input: the RIFF container is parsed here (integer header fields only), FLAC streams go through the native decoder
(``cw_flac_decode``, csrc/flac.cpp); sample scaling, mono mixdown and resampling run on the device (``cw_ingest``,
csrc/ingest.hip).  ``resample`` is ``torchaudio.functional.resample`` with its
defaults (TF/pipelines/automatic_speech_recognition.py:398-412), on the device as well.  Other containers raise like
the reference does when ffmpeg is missing.
"""
from __future__ import annotations

import struct
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


PCM_U8, PCM_S16, PCM_S24, PCM_S32, PCM_F32, PCM_F64 = range(6)
_NP_FMT = {np.dtype(np.uint8): PCM_U8, np.dtype(np.int16): PCM_S16, np.dtype(np.int32): PCM_S32,
           np.dtype(np.float32): PCM_F32, np.dtype(np.float64): PCM_F64}
_MALFORMED = ("Soundfile is either not in the correct format or is malformed. RIFF/WAV and FLAC input is decoded natively "
              "(the reference needs ffmpeg for anything else).")


def parse_wav(data: bytes):
    """RIFF/WAVE container -> (fmt code, channels, sample rate, n_frames, payload bytes).  Handles WAVE_FORMAT_PCM (1),
    IEEE_FLOAT (3) and EXTENSIBLE (0xFFFE, sub-format in the first two GUID bytes); chunks are word aligned."""
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(_MALFORMED)
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        body = pos + 8
        if cid == b"fmt ":
            if size < 16 or body + 16 > len(data):
                raise ValueError(_MALFORMED)
            tag, ch, sr, _, _, bits = struct.unpack_from("<HHIIHH", data, body)
            if tag == 0xFFFE and size >= 26 and body + 26 <= len(data):
                tag = struct.unpack_from("<H", data, body + 24)[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            payload = data[body: min(body + size, len(data))]     # streamed files may overstate the size
            break
        pos = body + size + (size & 1)
    if fmt is None or payload is None:
        raise ValueError(_MALFORMED)
    tag, ch, sr, bits = fmt
    code = {(1, 8): PCM_U8, (1, 16): PCM_S16, (1, 24): PCM_S24, (1, 32): PCM_S32, (3, 32): PCM_F32, (3, 64): PCM_F64}.get((tag, bits))
    if code is None or ch < 1 or sr < 1:
        raise ValueError(_MALFORMED)
    n_frames = len(payload) // (ch * (bits // 8))
    if n_frames < 1:
        raise ValueError(_MALFORMED)
    return code, ch, sr, n_frames, payload[: n_frames * ch * (bits // 8)]


import os as _os


# Upper bound on what one compressed file may expand to, in seconds (enforced inside the decoder's frame loop: a few hundred bytes
# of CONSTANT frames would otherwise expand without bound).  24 h: long-form transcription is a supported workload and the
# reference's ffmpeg_read has no cap.  CW_MAX_AUDIO_SECONDS overrides it, read at every call; a value that is not a positive
# integer is ignored.
MAX_DECODED_SECONDS = 24 * 3600


def _max_decoded_seconds() -> int:
    try:
        v = int(_os.environ.get("CW_MAX_AUDIO_SECONDS", ""))
        if v > 0:
            return v
    except ValueError:
        pass
    return int(MAX_DECODED_SECONDS)


def decode_flac(data: bytes):
    """FLAC bytes -> (int32 samples [frames, channels] left-justified to 32 bits, sample rate) through the native decoder
    (csrc/flac.cpp: RFC 9639 incl. CRC and MD5 verification; raises ValueError with the decoder's message)."""
    import ctypes as C
    from . import _native as N
    lib = N.load()
    buf = np.frombuffer(data, dtype=np.uint8)
    sr, ch, bps = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    total, n = C.c_int64(0), C.c_int64(0)
    ptr = buf.ctypes.data_as(C.c_void_p)
    if lib.cw_flac_info(ptr, len(buf), C.byref(sr), C.byref(ch), C.byref(bps), C.byref(total)) != 0:
        raise ValueError("malformed FLAC stream: " + (lib.cw_flac_last_error() or b"?").decode())
    # one decoding pass when STREAMINFO states the length (it is verified against the frames); a stream of unknown length
    # (total = 0) is sized by a first pass.  Either way the decoded audio is capped: a few hundred bytes of CONSTANT frames
    # would otherwise expand without bound.
    # the cap is in decoded sample frames at 16 kHz-equivalent duration, so a stream that declares a 655 kHz rate cannot buy 40x
    # the host memory: frames <= seconds * min(rate, 48 kHz)
    limit_s = _max_decoded_seconds()
    max_frames = limit_s * min(max(int(sr.value), 1), 48000)
    if total.value > max_frames:
        raise ValueError(f"FLAC stream declares {total.value} sample frames: more than the {max_frames} frames this path accepts "
                         f"(CW_MAX_AUDIO_SECONDS = {limit_s} s x min(sample rate {sr.value}, 48000) Hz; raise CW_MAX_AUDIO_SECONDS)")
    cap = int(total.value)
    if cap <= 0:
        if lib.cw_flac_decode(ptr, len(buf), None, max_frames, C.byref(n)) != 0:
            raise ValueError("malformed FLAC stream: " + (lib.cw_flac_last_error() or b"?").decode())
        cap = int(n.value)
        if cap > max_frames:
            raise ValueError(f"FLAC stream decodes to {cap} sample frames: more than the {max_frames} frames this path accepts "
                             f"(CW_MAX_AUDIO_SECONDS = {limit_s} s x min(sample rate {sr.value}, 48000) Hz; raise CW_MAX_AUDIO_SECONDS)")
    out = np.empty((cap, int(ch.value)), dtype=np.int32)
    if lib.cw_flac_decode(ptr, len(buf), out.ctypes.data_as(C.c_void_p), cap, C.byref(n)) != 0:
        raise ValueError("malformed FLAC stream: " + (lib.cw_flac_last_error() or b"?").decode())
    return out[: int(n.value)], int(sr.value)


def _need(engine):
    if engine is None or not hasattr(engine, "ingest"):
        raise RuntimeError("audio ingest runs on the device: pass the pipeline's Engine (there is no host fallback)")
    return engine


def resample(x: np.ndarray, sr_in: int, sr_out: int = SAMPLING_RATE, engine=None) -> np.ndarray:
    if int(sr_in) == int(sr_out):
        return x
    x = np.ascontiguousarray(x, dtype=np.float32)
    return _need(engine).ingest(x, PCM_F32, 1, len(x), int(sr_in), int(sr_out))


def decode_wav_bytes(data: bytes, sampling_rate: int = SAMPLING_RATE, engine=None, normalise: bool = False) -> np.ndarray:
    """WAV bytes -> mono float32 at ``sampling_rate``.  ``normalise=True`` applies REF/app.py:85-93
    ((y - mean) / std / 8 before resampling)."""
    if data[:4] == b"fLaC":                      # FLAC: container + entropy decoding on the host, the rest on the device
        pcm, sr = decode_flac(data)
        return _need(engine).ingest(pcm, PCM_S32, pcm.shape[1], pcm.shape[0], sr, sampling_rate, normalise=normalise)
    code, ch, sr, n_frames, payload = parse_wav(data)
    return _need(engine).ingest(payload, code, ch, n_frames, sr, sampling_rate, normalise=normalise)


def read_audio(path: str, sampling_rate: int = SAMPLING_RATE, engine=None) -> np.ndarray:
    with open(path, "rb") as f:
        return decode_wav_bytes(f.read(), sampling_rate, engine)
