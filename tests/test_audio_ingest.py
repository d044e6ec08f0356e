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


# ---------------------------------------------------------------------------------------------------------------------
# FLAC: native decoder (csrc/flac.cpp) against an independent bit-level writer (tests/flac_writer.py)
# ---------------------------------------------------------------------------------------------------------------------
def _flac_signal(rng, n, bps, nch, kind):
    t = np.arange(n)
    amp = (1 << (bps - 1)) - 1
    chans = []
    for c in range(nch):
        if kind == "tone":
            x = 0.6 * amp * np.sin(2 * np.pi * (0.01 + 0.003 * c) * t + c) + 0.02 * amp * rng.standard_normal(n)
        elif kind == "noise":
            x = rng.integers(-amp, amp, n).astype(np.float64)
        else:   # correlated stereo
            base = 0.5 * amp * np.sin(2 * np.pi * 0.007 * t)
            x = base + (0.05 * amp * rng.standard_normal(n) if c else 0)
        chans.append(np.clip(np.round(x), -amp - 1, amp).astype(np.int64))
    return chans


def _decode(data):
    from crisperwhisper_amd import audio
    pcm, sr = audio.decode_flac(data)
    return pcm, sr


def test_flac_decoder_every_subframe_type_and_stereo_mode_round_trips():
    """CONSTANT / VERBATIM / FIXED 0-4 / LPC (orders 1, 8, 32) subframes, Rice and Rice2 with partition orders 0-4 and escape
    partitions, wasted bits, the four channel assignments, 8 / 16 / 24 / 32-bit samples, STREAMINFO MD5 verified: the decoded
    samples equal the encoder's input exactly (left-justified to 32 bits)."""
    from tests import flac_writer as FW
    rng = np.random.default_rng(0)
    cases = 0
    for bps in (8, 16, 24, 32):
        for stereo in ("independent", "left_side", "side_right", "mid_side"):
            nch = 2
            n_blocks = [1024, 256, 4096, 192, 576, 100, 1000]
            total = sum(n_blocks)
            chans = _flac_signal(rng, total, bps if bps < 32 else 31, nch, "stereo" if stereo != "independent" else "tone")
            if bps == 32 and stereo != "independent":
                continue                                        # the side channel would need 33 bits of 32-bit input: own test below
            frames, pos = [], 0
            kinds = [("fixed", dict(order=2, method=0, porder=2)), ("lpc", None), ("verbatim", {}), ("fixed", dict(order=4, method=1, porder=0, escape_parts=(0,))),
                     ("fixed", dict(order=0, method=0, porder=3, escape_parts=(1, 5))), ("fixed", dict(order=1, method=1, porder=1)), ("fixed", dict(order=3, method=0, porder=3))]
            for nb, (kind, kw) in zip(n_blocks, kinds):
                plans = []
                for c in range(nch):
                    seg = chans[c][pos: pos + nb]
                    if kind == "lpc":
                        order = (1, 8, 32)[(cases + c) % 3]
                        # the subframe the writer emits is the (possibly decorrelated) channel: plan the LPC on the raw one, valid anyway
                        plans.append(dict(kind="lpc", order=order, method=c % 2, porder=2 if nb % 4 == 0 and nb // 4 >= order else 0,
                                          lpc=FW.lpc_plan(seg, order, prec=12 if bps <= 16 else 15)))
                    else:
                        plans.append(dict(kind=kind, **kw))
                frames.append(dict(n=nb, stereo=stereo, plans=plans))
                pos += nb
            data = FW.write_stream(chans, bps, 44100, frames)
            pcm, sr = _decode(data)
            assert sr == 44100 and pcm.shape == (total, nch)
            want = np.stack(chans, axis=1).astype(np.int64) << (32 - bps)
            assert np.array_equal(pcm.astype(np.int64), want), (bps, stereo)
            cases += 1
    assert cases >= 13


def test_flac_decoder_header_variants_constant_wasted_bits_and_unknown_length():
    """Mono / 3-channel streams, CONSTANT subframes, wasted bits, explicit 8- and 16-bit block sizes, sample rates given by the
    8-bit kHz / 16-bit Hz / 16-bit daHz codes or taken from STREAMINFO, variable-blocksize streams (sample numbers up to 5 coded
    bytes), a frame number that needs 3 coded bytes, extra metadata blocks in front of the audio, no MD5, unknown total length."""
    from tests import flac_writer as FW
    rng = np.random.default_rng(1)
    for sr in (16000, 37000, 12345, 655350, 48000):
        for nch in (1, 3):
            n_blocks = [200, 4096, 4096, 17]
            total = sum(n_blocks)
            chans = _flac_signal(rng, total, 16, nch, "tone")
            chans[0][:200] = 1234                                 # CONSTANT block
            for c in chans:
                c[200:4296] &= ~7                                 # 3 wasted bits in the second block
            frames, pos = [], 0
            for k, nb in enumerate(n_blocks):
                plans = []
                for c in range(nch):
                    if k == 0 and c == 0:
                        plans.append(dict(kind="constant"))
                    elif k == 1:
                        plans.append(dict(kind="fixed", order=2, method=0, porder=4, wasted=3))
                    else:
                        plans.append(dict(kind="fixed", order=c % 5, method=1, porder=0))
                frames.append(dict(n=nb, plans=plans, variable=(sr == 12345), use_streaminfo_sr=(sr == 48000 and k % 2 == 0),
                                   use_streaminfo_bps=(k == 2)))
                pos += nb
            extra = [(4, bytes(40)), (1, bytes(1000))]      # VORBIS_COMMENT-typed and PADDING blocks: skipped
            data = FW.write_stream(chans, 16, sr, frames, with_md5=(nch == 1), total_known=(nch == 3), extra_blocks=extra,
                                   first_number=70000)
            pcm, got_sr = _decode(data)
            assert got_sr == sr and pcm.shape == (total, nch)
            assert np.array_equal(pcm.astype(np.int64), np.stack(chans, axis=1).astype(np.int64) << 16), (sr, nch)


def test_flac_decoder_fails_loudly_on_corruption():
    """Flipped payload byte -> CRC-16, flipped header byte -> CRC-8 / sync, wrong MD5, truncation, not-FLAC: ValueError with the
    decoder's message, never wrong audio."""
    from tests import flac_writer as FW
    rng = np.random.default_rng(2)
    chans = _flac_signal(rng, 2048, 16, 2, "stereo")
    frames = [dict(n=1024, stereo="mid_side", plans=[dict(kind="fixed", order=2, method=0, porder=2)] * 2)] * 2
    good = FW.write_stream(chans, 16, 16000, frames)
    pcm, _ = _decode(good)
    assert pcm.shape == (2048, 2)
    audio_off = good.index(bytes([0xFF, 0xF8]))
    for mutate, msg in ((lambda b: b[:audio_off + 40] + bytes([b[audio_off + 40] ^ 0x10]) + b[audio_off + 41:], "CRC-16|residual|subframe|sync|bit depth"),
                        (lambda b: b[:audio_off + 2] + bytes([b[audio_off + 2] ^ 0x01]) + b[audio_off + 3:], "CRC-8|reserved|sync"),
                        (lambda b: b[:26] + bytes([b[26] ^ 0xFF]) + b[27:], "MD5"),
                        (lambda b: b[:len(b) - 300], "truncated|ends before|CRC|sync"),
                        (lambda b: b"RIFF" + b[4:], "fLaC")):
        with pytest.raises(ValueError, match=msg):
            _decode(mutate(good))

def test_flac_decompression_bomb_is_stopped_inside_the_frame_loop(monkeypatch):
    """A stream of CONSTANT subframes (14 bytes per 4096-sample block) of unknown total length: the decoder gives up as soon as
    the caller's bound is passed -- inside the frame loop, not after everything was materialised -- for the size query and for
    the decoding call alike; the same stream under a bound it fits in decodes."""
    import ctypes as C
    from crisperwhisper_amd import _native as N, audio
    from tests import flac_writer as FW
    n_blocks, bs = 40, 4096
    chans = [np.full(n_blocks * bs, 321, dtype=np.int64)]
    frames = [dict(n=bs, plans=[dict(kind="constant")]) for _ in range(n_blocks)]
    data = FW.write_stream(chans, 16, 16000, frames, with_md5=False, total_known=False)
    assert len(data) < 1000                                    # 160 k samples from < 1 kB
    lib = N.load()
    buf = np.frombuffer(data, dtype=np.uint8)
    n = C.c_int64(0)
    assert lib.cw_flac_decode(buf.ctypes.data_as(C.c_void_p), len(buf), None, 3 * bs, C.byref(n)) != 0
    assert b"more sample frames" in lib.cw_flac_last_error()
    out = np.empty((3 * bs, 1), np.int32)
    assert lib.cw_flac_decode(buf.ctypes.data_as(C.c_void_p), len(buf), out.ctypes.data_as(C.c_void_p), 3 * bs, C.byref(n)) != 0
    assert lib.cw_flac_decode(buf.ctypes.data_as(C.c_void_p), len(buf), None, n_blocks * bs, C.byref(n)) == 0 and n.value == n_blocks * bs
    monkeypatch.setattr(audio, "MAX_DECODED_SECONDS", 5)       # 80 k frames at 16 kHz < 160 k
    with pytest.raises(ValueError, match="more sample frames"):
        audio.decode_flac(data)
    monkeypatch.setattr(audio, "MAX_DECODED_SECONDS", 60)
    pcm, sr = audio.decode_flac(data)
    assert pcm.shape == (n_blocks * bs, 1) and sr == 16000 and int(pcm[0, 0]) == 321 << 16


@pytest.mark.gpu
def test_gpu_flac_path_equals_wav_path(eng, tmp_path):
    """A .flac file and the .wav file holding the same 16-bit stereo 44.1 kHz samples come out of `read_audio` (container decode
    on the host, sample scaling + mono mixdown + resampling on the device) bit-identical; 24-bit FLAC matches the f64 oracle."""
    from tests import flac_writer as FW
    rng = np.random.default_rng(4)
    n = 44100
    chans = _flac_signal(rng, n, 16, 2, "stereo")
    frames, pos = [], 0
    while pos < n:
        nb = min(4096, n - pos)
        frames.append(dict(n=nb, stereo="mid_side", plans=[dict(kind="fixed", order=2, method=0, porder=0)] * 2))
        pos += nb
    fl = tmp_path / "a.flac"; fl.write_bytes(FW.write_stream(chans, 16, 44100, frames))
    wv = tmp_path / "a.wav"; wv.write_bytes(_wav_bytes(np.stack(chans, axis=1).astype(np.int16), 44100))
    a = audio.read_audio(str(fl), 16000, eng)
    b = audio.read_audio(str(wv), 16000, eng)
    assert a.shape == b.shape == (16000,)
    assert np.array_equal(a, b)
    c24 = _flac_signal(rng, 8192, 24, 1, "tone")
    data = FW.write_stream(c24, 24, 16000, [dict(n=4096, plans=[dict(kind="fixed", order=3, method=1, porder=2)])] * 2)
    got = audio.decode_wav_bytes(data, 16000, eng)
    assert np.array_equal(got, (c24[0].astype(np.float64) / (1 << 23)).astype(np.float32))


@pytest.mark.parametrize("sr_in,sr_out", [(44100, 16000), (48000, 16000), (22050, 16000), (8000, 16000), (11025, 16000), (16000, 44100)])
def test_resampler_equals_the_closed_form_kernel_by_direct_convolution(sr_in, sr_out):
    """Independent route to the same numbers (no torchaudio exists offline, so the oracle stays "parity unpinned" against its
    OUTPUT; this pins it against its published DEFINITION): torchaudio's sinc_interp_hann resampler is, for every output sample m at
    input time tau = m * orig / new,
        y[m] = sum_n x[n] * (f0 / orig) * sinc(pi f0 (n - tau) / orig) * cos^2(pi f0 (n - tau) / (2 L orig)),   |f0 (n - tau) / orig| <= L,
    f0 = min(orig, new) * rolloff, L = lowpass_filter_width = 6 (functional.py: _get_sinc_resample_kernel /
    _apply_sinc_resample_kernel).  Evaluated here sample by sample in f64 from that formula -- no polyphase table, no strided
    convolution, no shared code with oracle/audio.py -- on random audio, and compared with the oracle's table-driven result."""
    import math
    from oracle import audio as OA
    rng = np.random.default_rng(sr_in + sr_out)
    n = 3000
    x = rng.standard_normal(n).astype(np.float32)
    g = math.gcd(sr_in, sr_out)
    orig, new = sr_in // g, sr_out // g
    f0 = min(orig, new) * 0.99
    L = 6.0
    n_out = -(-(n * new) // orig)
    width = math.ceil(L * orig / f0)                     # support in input samples (torchaudio zero-pads this many on both sides)
    y = np.zeros(n_out, np.float64)
    nn = np.arange(-width - 1, n + width + orig + 1, dtype=np.float64)
    xx = np.zeros(len(nn), np.float64)
    xx[width + 1: width + 1 + n] = x.astype(np.float64)
    for m in range(n_out):
        tau = m * orig / new
        # torchaudio evaluates the kernel on the integer grid idx in [-width, width + orig) around frame i = m // new:
        # taps cover input samples i*orig - width .. i*orig + width + orig - 1; the clamp makes the window zero outside |t| <= L
        i = m // new
        lo, hi = i * orig - width, i * orig + width + orig
        sel = (nn >= lo) & (nn < hi)
        t = (nn[sel] - tau) * f0 / orig
        tc = np.clip(t, -L, L)
        w = np.cos(tc * math.pi / L / 2.0) ** 2
        s = np.where(tc == 0.0, 1.0, np.sin(tc * math.pi) / np.where(tc == 0.0, 1.0, tc * math.pi))
        y[m] = float(np.sum(xx[sel] * s * w)) * f0 / orig
    got = OA.resample(x, sr_in, sr_out).astype(np.float64)
    assert len(got) == n_out
    # the oracle rounds its taps to f32 once (torchaudio's dtype=None branch): 6e-8 relative per tap
    assert np.abs(got - y).max() <= 2e-6 * max(1.0, np.abs(y).max()), np.abs(got - y).max()
