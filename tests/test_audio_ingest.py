"""Audio ingest: RIFF parsing on the host, sample decode / mono mixdown / resampling on the device.

CPU part: the oracle restatement of torchaudio's resampler against analytic properties (torchaudio itself is absent
offline: parity unpinned, see oracle/audio.py), and the RIFF parser against scipy's WAV writer/reader.
GPU part: cw_ingest / cw_resample_taps against the oracle."""
import io
import struct

import numpy as np
import pytest

from crisperwhisper_amd import audio
from oracle import audio as OA


def _sine(freq, sr, secs=1.0, phase=0.3):
    t = np.arange(int(sr * secs)) / sr
    return np.sin(2 * np.pi * freq * t + phase).astype(np.float32)


@pytest.mark.parametrize("sr_in", [8000, 11025, 22050, 32000, 44100, 48000])
def test_oracle_resampler_properties(sr_in):
    k, orig, new, width = OA.resample_taps(sr_in, 16000)
    assert k.shape == (new, 2 * width + orig) and k.dtype == np.float32
    assert np.abs(k.sum(1) - 1.0).max() < 2e-3                       # unit DC gain per phase
    x = _sine(440.0, sr_in)
    y = OA.resample(x, sr_in, 16000)
    assert len(y) == OA.resampled_length(len(x), sr_in, 16000) == int(np.ceil(len(x) * 16000 / sr_in))
    want = _sine(440.0, 16000)[: len(y)]
    assert np.abs(y[300:-300] - want[300:-300]).max() < 2e-3          # in-band tone keeps amplitude and phase
    if sr_in > 16000:                                                 # a tone above the new Nyquist is rejected
        z = OA.resample(_sine(0.45 * sr_in, sr_in), sr_in, 16000)
        assert np.abs(z[300:-300]).max() < 2e-2                      # >= 34 dB even just past the transition band (22.05 kHz case)
    assert OA.resample(x, 16000, 16000) is not None and np.array_equal(OA.resample(x, 16000, 16000), x)


def _wav_bytes(x, sr):
    from scipy.io import wavfile
    b = io.BytesIO()
    wavfile.write(b, sr, x)
    return b.getvalue()


@pytest.mark.parametrize("dtype,code", [(np.uint8, audio.PCM_U8), (np.int16, audio.PCM_S16), (np.int32, audio.PCM_S32),
                                        (np.float32, audio.PCM_F32), (np.float64, audio.PCM_F64)])
@pytest.mark.parametrize("channels", [1, 2])
def test_riff_parser_and_sample_decode_vs_scipy(dtype, code, channels):
    rng = np.random.default_rng(3)
    n = 1001
    if np.issubdtype(dtype, np.integer):
        info = np.iinfo(dtype)
        x = rng.integers(info.min, info.max, size=(n, channels), endpoint=True).astype(dtype)
    else:
        x = rng.standard_normal((n, channels)).astype(dtype)
    x = x[:, 0] if channels == 1 else x
    fmt, ch, sr, frames, payload = audio.parse_wav(_wav_bytes(x, 22050))
    assert (fmt, ch, sr, frames) == (code, channels, 22050, n)
    got = OA.pcm_to_mono(payload, fmt, ch)
    xf = x.astype(np.float32)
    if dtype == np.uint8:
        xf = (xf - 128.0) / 128.0
    elif np.issubdtype(dtype, np.integer):
        xf = xf / float(-np.iinfo(dtype).min)
    want = xf if channels == 1 else (xf[:, 0] + xf[:, 1]) / np.float32(2)
    assert np.array_equal(got, want.astype(np.float32))


def _s24_extensible_wav(vals, sr, channels):
    """24-bit WAVE_FORMAT_EXTENSIBLE with an odd-sized LIST chunk in front of the data (word alignment)."""
    raw = b"".join(struct.pack("<i", int(v))[:3] for v in vals)
    fmt = struct.pack("<HHIIHH", 0xFFFE, channels, sr, sr * channels * 3, channels * 3, 24) + struct.pack("<HHI", 22, 24, 0) + \
        struct.pack("<H", 1) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
    lst = b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\x00"
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + lst + b"data" + struct.pack("<I", len(raw)) + raw
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_riff_parser_s24_extensible_and_errors():
    vals = np.array([0, 1, -1, 8388607, -8388608, 123456, -654321, 42], dtype=np.int64)
    fmt, ch, sr, frames, payload = audio.parse_wav(_s24_extensible_wav(vals, 48000, 2))
    assert (fmt, ch, sr, frames) == (audio.PCM_S24, 2, 48000, 4)
    got = OA.pcm_to_mono(payload, fmt, ch)
    v = vals.astype(np.float32) / 8388608.0
    assert np.array_equal(got, ((v[0::2] + v[1::2]) / np.float32(2)).astype(np.float32))
    for bad in (b"", b"RIFFxxxxWAVE", b"OggS" + b"\0" * 64, _wav_bytes(np.zeros(4, np.int16), 16000)[:30]):
        with pytest.raises(ValueError):
            audio.parse_wav(bad)
    # ingest is device work: without an engine the shim refuses instead of falling back to the host
    with pytest.raises(RuntimeError):
        audio.decode_wav_bytes(_wav_bytes(np.zeros(16, np.int16), 8000))
    with pytest.raises(RuntimeError):
        audio.resample(np.zeros(16, np.float32), 8000, 16000)


# ---------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def eng():
    from crisperwhisper_amd.engine import Engine
    from tests import helpers as Hh
    g, v, W, spec = Hh.tiny_setup()
    e = Engine(spec, dtype="f32", max_batch=1)
    e.load_state_dict(W)
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sr_in,sr_out", [(44100, 16000), (48000, 16000), (8000, 16000), (22050, 16000), (11025, 16000),
                                          (32000, 16000), (16000, 8000), (16001, 16000)])
def test_gpu_resample_taps_and_output_vs_oracle(eng, sr_in, sr_out):
    import ctypes as C
    k, orig, new, width = OA.resample_taps(sr_in, sr_out)
    taps = np.zeros(k.size, np.float32)
    o, n, w = C.c_int32(), C.c_int32(), C.c_int32()
    assert eng.lib.cw_resample_taps(sr_in, sr_out, taps.ctypes.data_as(C.c_void_p), taps.size, C.byref(o), C.byref(n), C.byref(w)) == 0
    assert (o.value, n.value, w.value) == (orig, new, width)
    assert np.abs(taps.reshape(k.shape) - k).max() <= 2e-7           # both f64 -> f32; libm may differ in the last bit
    rng = np.random.default_rng(sr_in)
    n_in = sr_in // 2 + 37
    x = (rng.standard_normal(n_in) * 0.3).astype(np.float32) + _sine(300.0, sr_in, 1.0)[:n_in]
    got = eng.ingest(x, audio.PCM_F32, 1, len(x), sr_in, sr_out)
    want = OA.resample(x, sr_in, sr_out)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 5e-6 * max(1.0, np.abs(want).max())   # f32 FMA chain vs f64 accumulation


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["u8", "s16", "s24", "s32", "f32", "f64"])
@pytest.mark.parametrize("channels", [1, 2, 6])
def test_gpu_sample_decode_and_mixdown_bit_exact(eng, fmt, channels):
    rng = np.random.default_rng(11)
    n = 4099
    if fmt == "s24":
        vals = rng.integers(-8388608, 8388607, size=n * channels, endpoint=True)
        raw = b"".join(struct.pack("<i", int(v))[:3] for v in vals)
        code = audio.PCM_S24
    else:
        dt = {"u8": np.uint8, "s16": np.int16, "s32": np.int32, "f32": np.float32, "f64": np.float64}[fmt]
        if np.issubdtype(dt, np.integer):
            info = np.iinfo(dt)
            a = rng.integers(info.min, info.max, size=n * channels, endpoint=True).astype(dt)
        else:
            a = rng.standard_normal(n * channels).astype(dt)
        raw = a.tobytes()
        code = {"u8": audio.PCM_U8, "s16": audio.PCM_S16, "s32": audio.PCM_S32, "f32": audio.PCM_F32, "f64": audio.PCM_F64}[fmt]
    got = eng.ingest(raw, code, channels, n, 16000, 16000)
    assert np.array_equal(got, OA.pcm_to_mono(raw, code, channels))


@pytest.mark.gpu
def test_gpu_app_normalisation_and_wav_file_path(eng, tmp_path):
    from scipy.io import wavfile
    rng = np.random.default_rng(5)
    y = (rng.standard_normal((44100 * 2, 2)) * 3000 + 150).astype(np.int16)
    p = str(tmp_path / "clip.wav")
    wavfile.write(p, 44100, y)
    mono = OA.pcm_to_mono(y.tobytes(), OA.PCM_S16, 2)
    want = OA.resample(mono, 44100, 16000)
    got = audio.read_audio(p, 16000, eng)
    assert np.abs(got - want).max() < 5e-6
    # REF/app.py:85-93 on the mono signal: (y - mean) / std / 8, then resample
    want_n = OA.resample(OA.normalise(mono), 44100, 16000)
    got_n = audio.decode_wav_bytes(open(p, "rb").read(), 16000, eng, normalise=True)
    assert np.abs(got_n - want_n).max() < 2e-5
    with pytest.raises(Exception):
        eng.ingest(b"", audio.PCM_S16, 1, 0, 16000, 16000)


def test_riff_parser_never_crashes_on_corrupted_headers():
    """Fuzz: truncations and byte flips of valid WAV files either parse to a consistent description or raise ValueError
    (the reference's "Soundfile is either not in the correct format or is malformed" path) -- never another exception."""
    rng = np.random.default_rng(99)
    base = [_wav_bytes((rng.standard_normal((300, 2)) * 3000).astype(np.int16), 44100),
            _wav_bytes(rng.standard_normal(257).astype(np.float32), 8000),
            _s24_extensible_wav(rng.integers(-8388608, 8388607, size=64), 48000, 2)]
    n_ok = n_bad = 0
    for trial in range(600):
        b = bytearray(base[trial % 3])
        if trial % 2 == 0:
            b = b[: int(rng.integers(0, len(b)))]
        for _ in range(int(rng.integers(1, 4))):
            if len(b):
                b[int(rng.integers(0, min(len(b), 64)))] = int(rng.integers(0, 256))
        try:
            fmt, ch, sr, frames, payload = audio.parse_wav(bytes(b))
        except ValueError:
            n_bad += 1
            continue
        n_ok += 1
        bytes_per = {audio.PCM_U8: 1, audio.PCM_S16: 2, audio.PCM_S24: 3, audio.PCM_S32: 4, audio.PCM_F32: 4, audio.PCM_F64: 8}[fmt]
        assert ch >= 1 and sr >= 1 and frames >= 1 and len(payload) == frames * ch * bytes_per
    assert n_ok > 0 and n_bad > 0


def test_resampler_rejects_absurd_rate_pairs_without_allocating():
    """A corrupted header can claim any sampling rate: coprime MHz-range rates would need a multi-GB polyphase table;
    the library answers with an error code (checked through the context-free cw_resample_taps entry point)."""
    import ctypes as C
    from crisperwhisper_amd import _native
    lib = _native.load()
    o, n, w = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.cw_resample_taps(2147483647, 16000, None, 0, C.byref(o), C.byref(n), C.byref(w)) != 0
    assert lib.cw_resample_taps(1000003, 999983, None, 0, C.byref(o), C.byref(n), C.byref(w)) != 0
    assert lib.cw_resample_taps(44100, 16000, None, 0, C.byref(o), C.byref(n), C.byref(w)) == 0 and (o.value, n.value, w.value) == (441, 160, 17)
    assert lib.cw_resampled_length(441000, 44100, 16000) == 160000
