"""Thin object wrapper over the C ABI: one ``Engine`` = one ``cw_ctx`` on one GPU.

Host side of seam 2 of SURVEY.md section 8b: owns the context, uploads weights, and exposes the device
stages (mel / encode / decode / token timestamps) with numpy host buffers.  No arithmetic of the
hot path happens in Python.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _native as N


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i32(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=np.int32))


@dataclasses.dataclass
class ModelSpec:
    """The fields of WhisperConfig / generation_config the path reads."""
    d_model: int
    n_heads: int
    ffn_dim: int
    enc_layers: int
    dec_layers: int
    n_mels: int
    vocab_size: int
    max_target_positions: int = 448
    median_filter_width: int = 7
    alignment_heads: Sequence[Sequence[int]] = ()
    eos_token_id: int = 0
    pad_token_id: int = 0
    decoder_start_token_id: int = 0
    no_timestamps_token_id: int = 0
    max_initial_timestamp_index: Optional[int] = 50
    suppress_tokens: Sequence[int] = ()
    begin_suppress_tokens: Sequence[int] = ()
    lang_to_id: Dict[str, int] = dataclasses.field(default_factory=dict)
    task_to_id: Dict[str, int] = dataclasses.field(default_factory=dict)
    max_length: int = 448
    # generation_config defaults that seed the decoder prompt (generation_whisper.py:1488-1525)
    forced_decoder_ids: Optional[Sequence[Sequence[Optional[int]]]] = None
    language: Optional[str] = None
    task: Optional[str] = None

    @property
    def timestamp_begin(self) -> int:
        return self.no_timestamps_token_id + 1


class EngineError(RuntimeError):
    pass


class Engine:
    def __init__(self, spec: ModelSpec, dtype: str = "bf16", max_batch: int = 16, device: int = 0,
                 cross_kv_dtype: Optional[str] = None, encoder_gemm_dtype: Optional[str] = None):
        """``dtype``: "f32" (parity engine), "bf16" or "f16" (the 16-bit MFMA engine in bfloat16 / IEEE binary16).
        ``cross_kv_dtype="fp8"`` (16-bit engines only): the decode step streams an OCP e4m3 copy of the cross-attention
        cache, half the bytes of its dominant stream -- an accuracy-gated performance mode, not the parity path.
        ``encoder_gemm_dtype="fp8"`` (16-bit engines only; BASELINE configs[3]): the encoder's qkv / fc1 / fc2 projections and
        the cross-K/V projection run as e4m3 x e4m3 MFMA GEMMs with row-wise scales (weights quantised once after loading,
        activations by the LayerNorm that produces them) -- likewise accuracy-gated, not the parity path.  ``"fp8:fc1"`` (any subset of
        qkv / fc1 / fc2 / cross_kv joined by "+") quantises only those; "fp8:fc1" together with ``cross_kv_dtype="fp8"`` is the
        largest subset that reproduces every reference clip of the batch-64 workload (profiles/r04_fp8_sweep.txt)."""
        self.lib = N.load()
        self.spec = spec
        self.dtype = dtype
        self.max_batch = int(max_batch)
        if not spec.alignment_heads:
            raise ValueError("Model generation config has no `alignment_heads`, token-level timestamps not available.")
        al = _i32([h[0] for h in spec.alignment_heads])
        ah = _i32([h[1] for h in spec.alignment_heads])
        desc = N.ModelDesc(
            d_model=spec.d_model, n_heads=spec.n_heads, ffn_dim=spec.ffn_dim, enc_layers=spec.enc_layers,
            dec_layers=spec.dec_layers, n_mels=spec.n_mels, vocab_size=spec.vocab_size,
            max_target_positions=spec.max_target_positions, median_filter_width=spec.median_filter_width,
            dtype={"f32": N.CW_DTYPE_F32, "fp32": N.CW_DTYPE_F32, "bf16": N.CW_DTYPE_BF16, "f16": N.CW_DTYPE_F16, "fp16": N.CW_DTYPE_F16}[dtype],
            max_batch=self.max_batch, n_align=len(al),
            align_layers=al.ctypes.data_as(C.POINTER(C.c_int32)), align_heads=ah.ctypes.data_as(C.POINTER(C.c_int32)))
        self.ctx = self.lib.cw_create(C.byref(desc), int(device))
        if not self.ctx:
            raise EngineError("cw_create failed: " + (self.lib.cw_last_error(None) or b"?").decode())
        sup, bsup = _i32(list(spec.suppress_tokens)), _i32(list(spec.begin_suppress_tokens))
        cfg = N.GenCfg(
            eos_token_id=spec.eos_token_id, pad_token_id=spec.pad_token_id,
            no_timestamps_token_id=spec.no_timestamps_token_id,
            max_initial_timestamp_index=-1 if spec.max_initial_timestamp_index is None else spec.max_initial_timestamp_index,
            suppress_tokens=sup.ctypes.data_as(C.POINTER(C.c_int32)), n_suppress=len(sup),
            begin_suppress_tokens=bsup.ctypes.data_as(C.POINTER(C.c_int32)), n_begin_suppress=len(bsup))
        self._chk(self.lib.cw_set_generation(self.ctx, C.byref(cfg)))
        self._capture = None
        if cross_kv_dtype not in (None, "bf16", "f32", "fp8"):
            raise ValueError(f"cross_kv_dtype must be None or 'fp8', got {cross_kv_dtype!r}")
        if cross_kv_dtype == "fp8":
            self._chk(self.lib.cw_set_option(self.ctx, b"cross_kv_fp8", 1))
        sub = encoder_gemm_dtype[4:] if isinstance(encoder_gemm_dtype, str) and encoder_gemm_dtype.startswith("fp8:") else None
        if sub is not None and (not sub or any(t not in ("qkv", "fc1", "fc2", "cross_kv") for t in sub.split("+"))):
            raise ValueError(f"encoder_gemm_dtype {encoder_gemm_dtype!r}: the subset after 'fp8:' is qkv / fc1 / fc2 / cross_kv joined by '+'")
        if sub is None and encoder_gemm_dtype not in (None, "bf16", "f16", "f32", "fp8"):
            raise ValueError(f"encoder_gemm_dtype must be None, 'fp8' or 'fp8:<subset>', got {encoder_gemm_dtype!r}")
        self._enc_fp8 = sub if sub is not None else (encoder_gemm_dtype == "fp8")

    # ------------------------------------------------------------------
    def _chk(self, rc: int):
        if rc != 0:
            raise EngineError(f"native call failed ({rc}): " + (self.lib.cw_last_error(self.ctx) or b"?").decode())

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.cw_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._chk(self.lib.cw_sync(self.ctx))

    # ------------------------------------------------------------------ weights
    def load_tensor(self, name: str, array: np.ndarray):
        a = np.ascontiguousarray(array, dtype=np.float32)
        shape = (C.c_int64 * a.ndim)(*a.shape)
        self._chk(self.lib.cw_load_tensor(self.ctx, name.encode(), _ptr(a), shape, a.ndim))

    def load_state_dict(self, weights: Dict[str, np.ndarray]):
        """Uploads every tensor, then fails loudly (naming the absent tensors) if the checkpoint was incomplete."""
        for k, v in weights.items():
            self.load_tensor(k, v)
        self.check_weights()
        if getattr(self, "_enc_fp8", False):
            self.set_encoder_gemm_fp8(self._enc_fp8)

    _ENC8_BITS = {"qkv": 1, "fc1": 2, "fc2": 4, "cross_kv": 8}

    def set_encoder_gemm_fp8(self, on, mask: Optional[int] = None):
        """(Re)build the e4m3 copies of the resident encoder / cross-K/V weights and switch the encoder GEMMs to them, or back.
        ``on``: True (or the legacy int 1) = every GEMM, False / 0 = none; a str of names out of "qkv", "fc1", "fc2", "cross_kv"
        joined by "+" = that subset.  ``mask`` (keyword) gives the subset as bits instead: 1 q/k/v, 2 fc1, 4 fc2, 8 cross-K/V
        projection.  "fc1" (together with the e4m3 cross-attention cache) is the largest subset that reproduces every reference
        clip of the batch-64 workload (profiles/r04_fp8_sweep.txt: 64 / 64; "fc1+fc2" is at 62-63 / 64)."""
        if mask is not None:
            if isinstance(mask, bool) or not 0 <= int(mask) <= 15:
                raise ValueError(f"mask must be a bit mask in 0..15 (1 q/k/v, 2 fc1, 4 fc2, 8 cross-K/V), not {mask!r}")
            value = 16 + int(mask) if int(mask) > 0 else 0
        elif isinstance(on, str):
            names = [t for t in on.split("+") if t]
            unknown = [t for t in names if t not in self._ENC8_BITS]
            if unknown or not names:
                raise ValueError(f"unknown encoder GEMM name(s) {unknown or on!r}: choose from {sorted(self._ENC8_BITS)} joined by '+'")
            value = 16 + sum(self._ENC8_BITS[t] for t in set(names))
        elif isinstance(on, (bool, np.bool_)) or on in (0, 1):
            value = 1 if on else 0
        else:
            raise ValueError(f"set_encoder_gemm_fp8({on!r}): pass True / False, a '+'-joined name list, or mask=<bits>")
        self._chk(self.lib.cw_set_option(self.ctx, b"encoder_gemm_fp8", value))

    def check_weights(self):
        self._chk(self.lib.cw_check_weights(self.ctx))

    # ------------------------------------------------------------------ stages
    def mel(self, clips: List[np.ndarray], return_features: bool = False):
        """clips: list of 1-D float32 arrays (each <= 30 s).  Returns (features or None, n_frames)."""
        B = len(clips)
        ns = _i32([len(c) for c in clips])
        pcm = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.float32) for c in clips])) if B else np.zeros(0, np.float32)
        feats = np.empty((B, self.spec.n_mels, N.N_FRAMES), dtype=np.float32) if return_features else None
        nf = np.zeros(B, dtype=np.int32)
        self._chk(self.lib.cw_mel(self.ctx, _ptr(pcm), B, _ptr(ns), _ptr(feats), _ptr(nf)))
        return feats, nf

    def upload_pcm(self, clips: List[np.ndarray]):
        ns = _i32([len(c) for c in clips])
        pcm = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.float32) for c in clips]))
        self._chk(self.lib.cw_upload_pcm(self.ctx, _ptr(pcm), len(clips), _ptr(ns)))
        return (ns + 159) // 160

    def mel_resident(self, B: int):
        self._chk(self.lib.cw_mel_resident(self.ctx, B))

    def set_features(self, feats: np.ndarray):
        f = np.ascontiguousarray(feats, dtype=np.float32)
        self._chk(self.lib.cw_set_features(self.ctx, _ptr(f), f.shape[0]))

    def encode(self, item, seek, n_frames):
        item, seek, n_frames = _i32(item), _i32(seek), _i32(n_frames)
        self._chk(self.lib.cw_encode(self.ctx, len(item), _ptr(item), _ptr(seek), _ptr(n_frames)))

    def encoder_output(self, nb: int) -> np.ndarray:
        out = np.empty((nb, N.N_CTX, self.spec.d_model), dtype=np.float32)
        self._chk(self.lib.cw_get_encoder_output(self.ctx, _ptr(out), nb))
        return out

    def decode(self, prompt: np.ndarray, max_length: int, min_new_tokens: int = 0,
               forced: Optional[np.ndarray] = None, want_argmax: bool = False):
        prompt = _i32(prompt)
        nb, n_prompt = prompt.shape
        tgt = self.spec.max_target_positions
        seqs = np.zeros((nb, tgt), dtype=np.int32)
        lens = np.zeros(nb, dtype=np.int32)
        amax = np.zeros((nb, tgt), dtype=np.int32) if want_argmax else None
        f = None
        if forced is not None:
            f = np.full((nb, tgt), -1, dtype=np.int32)
            f[:, :forced.shape[1]] = forced
        self._chk(self.lib.cw_decode(self.ctx, nb, _ptr(prompt), n_prompt, int(max_length), int(min_new_tokens),
                                     _ptr(f), _ptr(seqs), _ptr(lens), _ptr(amax)))
        return seqs, lens, amax

    def set_thresholds(self, logprob_threshold: Optional[float] = None, no_speech_threshold: Optional[float] = None):
        """Deterministic half of generate_with_fallback (``cw_set_thresholds``); None = unset."""
        nan = float("nan")
        self._chk(self.lib.cw_set_thresholds(self.ctx, nan if logprob_threshold is None else float(logprob_threshold),
                                             nan if no_speech_threshold is None else float(no_speech_threshold)))

    def no_speech_probs(self, nb: int, sot: int) -> np.ndarray:
        out = np.zeros(nb, np.float32)
        self._chk(self.lib.cw_no_speech_probs(self.ctx, nb, int(sot), _ptr(out)))
        return out

    def avg_logprobs(self, nb: int) -> np.ndarray:
        out = np.zeros(nb, np.float32)
        self._chk(self.lib.cw_get_avg_logprobs(self.ctx, _ptr(out), nb))
        return out

    def last_logits(self, nb: int) -> np.ndarray:
        out = np.empty((nb, self.spec.vocab_size), dtype=np.float32)
        self._chk(self.lib.cw_get_logits(self.ctx, _ptr(out), nb))
        return out

    def capture_logits(self, nb: int, max_steps: int) -> np.ndarray:
        self._capture = np.zeros((max_steps, nb, self.spec.vocab_size), dtype=np.float32)
        self._chk(self.lib.cw_set_logits_capture(self.ctx, _ptr(self._capture), max_steps))
        return self._capture

    def stop_capture(self):
        self._chk(self.lib.cw_set_logits_capture(self.ctx, None, 0))
        self._capture = None

    def alignment(self, nb: int, L: int) -> np.ndarray:
        out = np.empty((nb, len(self.spec.alignment_heads), L, N.N_CTX), dtype=np.float32)
        self._chk(self.lib.cw_get_alignment(self.ctx, _ptr(out), nb, L))
        return out

    def ingest(self, raw, fmt: int, channels: int, n_frames: int, sr_in: int, sr_out: int = 16000,
               normalise: bool = False) -> np.ndarray:
        """cw_ingest: interleaved sample frames (bytes or a C-contiguous array) -> mono float32 at ``sr_out``."""
        buf = np.frombuffer(raw, dtype=np.uint8) if isinstance(raw, (bytes, bytearray, memoryview)) else np.ascontiguousarray(raw)
        n_out = int(self.lib.cw_resampled_length(int(n_frames), int(sr_in), int(sr_out)))
        out = np.empty(n_out, dtype=np.float32)
        self._chk(self.lib.cw_ingest(self.ctx, buf.ctypes.data_as(C.c_void_p), int(fmt), int(channels), int(n_frames),
                                     int(sr_in), int(sr_out), 1 if normalise else 0, _ptr(out)))
        return out

    def transcribe(self, nb: int, num_frames, *, sot: int, language_token: int = -1, task_token: int = -1,
                   max_new_tokens: int = -1, min_new_tokens: int = 0, max_length: int = 448, lang_ids=None):
        """Native seek loop (cw_transcribe) over the nb resident feature items: returns (tokens, timestamps, passes),
        the per-item concatenated segment tokens (int64) and absolute token timestamps (float32)."""
        nf = _i32(num_frames)
        lids = _i32(lang_ids if lang_ids is not None else [])
        cfg = N.TranscribeCfg(sot, language_token, task_token, max_new_tokens, min_new_tokens, max_length,
                              lids.ctypes.data_as(C.POINTER(C.c_int32)), len(lids))
        # a 30 s window can be re-decoded at most once per 0.02 s of progress; 4 full-length passes is far above
        # what the seek loop can emit before running out of frames with real timestamps, and the call fails loudly
        # (never truncates) if an item exceeds it
        cap = 4 * self.spec.max_target_positions
        while True:
            toks = np.zeros((nb, cap), dtype=np.int32)
            ts = np.zeros((nb, cap), dtype=np.float32)
            lens = np.zeros(nb, dtype=np.int32)
            passes = C.c_int32(0)
            rc = self.lib.cw_transcribe(self.ctx, nb, _ptr(nf), C.byref(cfg), _ptr(toks), _ptr(ts), _ptr(lens), cap,
                                        C.byref(passes))
            if rc != 0 and b"capacity" in (self.lib.cw_last_error(self.ctx) or b"") and cap < (1 << 20):
                cap *= 4
                continue
            self._chk(rc)
            break
        return ([toks[i, :lens[i]].astype(np.int64) for i in range(nb)], [ts[i, :lens[i]].copy() for i in range(nb)],
                int(passes.value))

    # ------------------------------------------------------------------ beam search (device half; host half: generation.beam_search)
    def beam_begin(self, prompt: np.ndarray, num_beams: int, max_length: int, min_new_tokens: int = 0):
        prompt = _i32(prompt)
        n_items, n_prompt = prompt.shape
        self._beam_rows = n_items * int(num_beams)
        self._chk(self.lib.cw_beam_begin(self.ctx, n_items, int(num_beams), _ptr(prompt), n_prompt, int(max_length),
                                         int(min_new_tokens)))

    def beam_step(self, n_cand: int):
        """-> (log-probabilities [rows, n_cand] float32 best first, tokens [rows, n_cand] int32; -inf / -1 padded)."""
        vals = np.empty((self._beam_rows, n_cand), dtype=np.float32)
        toks = np.empty((self._beam_rows, n_cand), dtype=np.int32)
        self._chk(self.lib.cw_beam_step(self.ctx, int(n_cand), _ptr(vals), _ptr(toks)))
        return vals, toks

    def beam_advance(self, parent, token):
        parent, token = _i32(parent), _i32(token)
        self._chk(self.lib.cw_beam_advance(self.ctx, _ptr(parent), _ptr(token)))

    def beam_finish(self, row_of_pos: np.ndarray):
        r = _i32(row_of_pos)
        self._chk(self.lib.cw_beam_finish(self.ctx, r.shape[0], r.shape[1], _ptr(r)))

    def token_timestamps(self, nb: int, L: int, n_prompt: int, num_frames) -> np.ndarray:
        nf = _i32(num_frames)
        out = np.zeros((nb, L + 1), dtype=np.float32)
        self._chk(self.lib.cw_token_timestamps(self.ctx, nb, L, n_prompt, _ptr(nf), _ptr(out)))
        return out

    # ------------------------------------------------------------------ stand-alone kernels
    def align_matrix(self, attn: np.ndarray, n_cols, width: int) -> np.ndarray:
        a = np.ascontiguousarray(attn, dtype=np.float32)
        B, Ha, Nn, M = a.shape
        nc = _i32(n_cols)
        out = np.zeros((B, Nn, M), dtype=np.float32)
        self._chk(self.lib.cw_align_matrix(self.ctx, _ptr(a), B, Ha, Nn, M, _ptr(nc), width, _ptr(out)))
        return out

    def dtw(self, mat: np.ndarray):
        m = np.ascontiguousarray(mat, dtype=np.float32)
        Nn, M = m.shape
        ti = np.zeros(Nn + M + 2, dtype=np.int32)
        tj = np.zeros(Nn + M + 2, dtype=np.int32)
        n = C.c_int32(0)
        self._chk(self.lib.cw_dtw(self.ctx, _ptr(m), Nn, M, _ptr(ti), _ptr(tj), C.byref(n)))
        return ti[:n.value].copy(), tj[:n.value].copy()

    def adjust_pauses(self, start: np.ndarray, end: np.ndarray, thr: float):
        s = np.ascontiguousarray(start, dtype=np.float64).copy()
        e = np.ascontiguousarray(end, dtype=np.float64).copy()
        self._chk(self.lib.cw_adjust_pauses(self.ctx, _ptr(s), _ptr(e), len(s), float(thr)))
        return s, e

    def test_gemm(self, A, W, bias=None, gelu=False):
        A = np.ascontiguousarray(A, np.float32); W = np.ascontiguousarray(W, np.float32)
        b = None if bias is None else np.ascontiguousarray(bias, np.float32)
        out = np.zeros((A.shape[0], W.shape[0]), np.float32)
        self._chk(self.lib.cw_test_gemm(self.ctx, A.shape[0], W.shape[0], A.shape[1], _ptr(A), _ptr(W), _ptr(b), int(gelu), _ptr(out)))
        return out

    def test_gemm_fp8(self, A, W, bias=None, gelu=False):
        """A W^T (+ bias, optional GELU) through the e4m3 GEMM of the opt-in fp8 encoder mode (row-wise scales of both operands)."""
        A = np.ascontiguousarray(A, np.float32); W = np.ascontiguousarray(W, np.float32)
        b = None if bias is None else np.ascontiguousarray(bias, np.float32)
        out = np.zeros((A.shape[0], W.shape[0]), np.float32)
        self._chk(self.lib.cw_test_gemm_fp8(self.ctx, A.shape[0], W.shape[0], A.shape[1], _ptr(A), _ptr(W), _ptr(b), int(gelu), _ptr(out)))
        return out

    def test_gemv(self, x, W, bias=None, ln=None, gelu=False):
        x = np.ascontiguousarray(x, np.float32); W = np.ascontiguousarray(W, np.float32)
        b = None if bias is None else np.ascontiguousarray(bias, np.float32)
        g = be = None
        if ln is not None:
            g, be = (np.ascontiguousarray(t, np.float32) for t in ln)
        out = np.zeros((x.shape[0], W.shape[0]), np.float32)
        self._chk(self.lib.cw_test_gemv(self.ctx, x.shape[0], W.shape[0], x.shape[1], _ptr(x), _ptr(W), _ptr(b),
                                        _ptr(g), _ptr(be), int(gelu), _ptr(out)))
        return out

    def test_skinny(self, mode, x, W, bias=None, out0=None, nks=0, reps=0):
        """One skinny-M decoder projection (cw_test_skinny, 17..64 rows).  mode 0: LayerNorm (no affine) + projection, 1: + GELU
        (16-bit result), 2: out0 + x16 W^T + bias on the residual grid.  Returns (out, (gemm_us, finish_us))."""
        x = np.ascontiguousarray(x, np.float32); W = np.ascontiguousarray(W, np.float32)
        b = None if bias is None else np.ascontiguousarray(bias, np.float32)
        out = np.zeros((x.shape[0], W.shape[0]), np.float32) if out0 is None else np.ascontiguousarray(out0, np.float32).copy()
        us = np.zeros(2, np.float32)
        self._chk(self.lib.cw_test_skinny(self.ctx, int(mode), x.shape[0], W.shape[0], x.shape[1], _ptr(x), _ptr(W), _ptr(b),
                                          int(nks), int(reps), _ptr(out), _ptr(us)))
        return out, (float(us[0]), float(us[1]))

    def test_attention(self, q, k, v):
        q, k, v = (np.ascontiguousarray(t, np.float32) for t in (q, k, v))
        B, H, S, _ = q.shape
        out = np.zeros((B, S, H * 64), np.float32)
        self._chk(self.lib.cw_test_attention(self.ctx, B, H, S, _ptr(q), _ptr(k), _ptr(v), _ptr(out)))
        return out

    def test_cross_attention(self, q, k, v, kv_div=1, align_head=0):
        """One launch of the key-split cross-attention decode kernel (cw_test_cross_attention): q [B][H][64] pre-scaled,
        k / v [B / kv_div][H][S][64].  Returns (out [B][H*64], align [B][S]) with the six splits combined on the host."""
        q, k, v = (np.ascontiguousarray(t, np.float32) for t in (q, k, v))
        B, H, _ = q.shape
        S = k.shape[2]
        NS = 6
        po = np.zeros((NS, B, H * 64), np.float32); ml = np.zeros((B, H, NS, 2), np.float32)
        al = np.zeros((B, S), np.float32); aml = np.zeros((B, NS, 2), np.float32)
        self._chk(self.lib.cw_test_cross_attention(self.ctx, B, H, S, int(kv_div), _ptr(q), _ptr(k), _ptr(v), int(align_head),
                                                   _ptr(po), _ptr(ml), _ptr(al), _ptr(aml)))
        m = ml[..., 0].astype(np.float64); l = ml[..., 1].astype(np.float64)
        M = m.max(-1, keepdims=True)
        w = np.exp(m - M)                                            # [B][H][NS]
        o = (po.astype(np.float64).reshape(NS, B, H, 64) * w.transpose(2, 0, 1)[..., None]).sum(0) / (l * w).sum(-1)[..., None]
        per = (S + NS - 1) // NS
        am = aml[..., 0].astype(np.float64); a_l = aml[..., 1].astype(np.float64)
        AM = am.max(-1, keepdims=True); aw = np.exp(am - AM)
        scale = np.repeat(aw, per, axis=1)[:, :S] / (a_l * aw).sum(-1, keepdims=True)
        return o.reshape(B, H * 64), al.astype(np.float64) * scale

    def test_sample(self, logits: np.ndarray, ids: np.ndarray, n_prompt: int, min_new_tokens: int = 0,
                    max_length: Optional[int] = None) -> np.ndarray:
        """One launch of the fused logits processors + greedy choice on caller rows (cw_test_sample)."""
        lg = np.ascontiguousarray(logits, np.float32)
        ids = _i32(ids)
        nb, t = ids.shape
        out = np.zeros(nb, np.int32)
        self._chk(self.lib.cw_test_sample(self.ctx, nb, _ptr(lg), _ptr(ids), t, int(n_prompt), int(min_new_tokens),
                                          int(max_length or self.spec.max_target_positions), _ptr(out)))
        return out

    # ------------------------------------------------------------------ measurement
    def stage_times(self, reset: bool = False):
        ms = np.zeros(len(N.STAGES), np.float32)
        calls = np.zeros(len(N.STAGES), np.int32)
        self._chk(self.lib.cw_stage_times(self.ctx, _ptr(ms), _ptr(calls), int(reset)))
        return {s: (float(ms[i]), int(calls[i])) for i, s in enumerate(N.STAGES)}

    def time_kernel(self, which: int, nb: int, iters: int):
        ms = C.c_float(0.0)
        by = C.c_double(0.0)
        self._chk(self.lib.cw_time_kernel(self.ctx, which, nb, iters, C.byref(ms), C.byref(by)))
        return ms.value, by.value

    def time_decode_stages(self, nb: int, iters: int):
        """Every launch of the decoder layer as the decode step issues it for `nb` greedy rows, timed one at a time (HIP events on
        the engine's stream, layers cycled): [{"stage", "kernel", "avg_ms", "algo_bytes"}], plus the whole layer as stage -1."""
        out = []
        ms, by, kind, ns = C.c_float(0.0), C.c_double(0.0), C.c_int32(0), C.c_int32(0)
        self._chk(self.lib.cw_time_decode_stage(self.ctx, nb, -1, iters, C.byref(ms), C.byref(by), C.byref(kind), C.byref(ns)))
        out.append({"stage": -1, "kernel": "whole decoder layer (all launches)", "avg_ms": ms.value, "algo_bytes": by.value})
        for s in range(ns.value):
            self._chk(self.lib.cw_time_decode_stage(self.ctx, nb, s, iters, C.byref(ms), C.byref(by), C.byref(kind), C.byref(ns)))
            out.append({"stage": s, "kernel": self.lib.cw_decode_stage_name(kind.value).decode(), "avg_ms": ms.value, "algo_bytes": by.value,
                        "launches": int(self.lib.cw_decode_stage_launches(self.ctx, s))})
        return out
