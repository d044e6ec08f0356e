"""ctypes binding of libcrisperwhisper.so (include/crisperwhisper.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C crisperwhisper_amd/csrc``.
There is no CPU fallback: if the library is missing or fails to load, every use raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CW_LIB_PATH: a development build of the same library (A/B of compile-time variants, e.g. -DCW_PHASE_TIMING); never a fallback
LIB_PATH = os.environ.get("CW_LIB_PATH") or os.path.join(_HERE, "libcrisperwhisper.so")

CW_DTYPE_F32, CW_DTYPE_BF16, CW_DTYPE_F16 = 0, 1, 2
N_SAMPLES, N_FRAMES, N_CTX = 480000, 3000, 1500
STAGES = ("mel", "encoder", "cross_kv", "decode", "timestamps")


class NativeLibraryError(RuntimeError):
    pass


class ModelDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "d_model", "n_heads", "ffn_dim", "enc_layers", "dec_layers", "n_mels", "vocab_size",
        "max_target_positions", "median_filter_width", "dtype", "max_batch", "n_align")] + [
        ("align_layers", C.POINTER(C.c_int32)), ("align_heads", C.POINTER(C.c_int32))]


class GenCfg(C.Structure):
    _fields_ = [("eos_token_id", C.c_int32), ("pad_token_id", C.c_int32),
                ("no_timestamps_token_id", C.c_int32), ("max_initial_timestamp_index", C.c_int32),
                ("suppress_tokens", C.POINTER(C.c_int32)), ("n_suppress", C.c_int32),
                ("begin_suppress_tokens", C.POINTER(C.c_int32)), ("n_begin_suppress", C.c_int32)]


class TranscribeCfg(C.Structure):
    _fields_ = [("sot_token", C.c_int32), ("language_token", C.c_int32), ("task_token", C.c_int32),
                ("max_new_tokens", C.c_int32), ("min_new_tokens", C.c_int32), ("max_length", C.c_int32),
                ("lang_ids", C.POINTER(C.c_int32)), ("n_lang_ids", C.c_int32)]


_P = C.c_void_p
_I = C.c_int32
_SIGS = {
    "cw_abi_version": (_I, []),
    "cw_create": (_P, [C.POINTER(ModelDesc), _I]),
    "cw_destroy": (None, [_P]),
    "cw_last_error": (C.c_char_p, [_P]),
    "cw_sync": (_I, [_P]),
    "cw_load_tensor": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I]),
    "cw_check_weights": (_I, [_P]),
    "cw_set_generation": (_I, [_P, C.POINTER(GenCfg)]),
    "cw_mel": (_I, [_P, _P, _I, _P, _P, _P]),
    "cw_upload_pcm": (_I, [_P, _P, _I, _P]),
    "cw_mel_resident": (_I, [_P, _I]),
    "cw_set_features": (_I, [_P, _P, _I]),
    "cw_encode": (_I, [_P, _I, _P, _P, _P]),
    "cw_get_encoder_output": (_I, [_P, _P, _I]),
    "cw_decode": (_I, [_P, _I, _P, _I, _I, _I, _P, _P, _P, _P]),
    "cw_set_thresholds": (_I, [_P, C.c_float, C.c_float]),
    "cw_no_speech_probs": (_I, [_P, _I, _I, _P]),
    "cw_get_avg_logprobs": (_I, [_P, _P, _I]),
    "cw_get_logits": (_I, [_P, _P, _I]),
    "cw_set_logits_capture": (_I, [_P, _P, _I]),
    "cw_get_alignment": (_I, [_P, _P, _I, _I]),
    "cw_resampled_length": (C.c_int64, [C.c_int64, _I, _I]),
    "cw_ingest": (_I, [_P, _P, _I, _I, C.c_int64, _I, _I, _I, _P]),
    "cw_resample_taps": (_I, [_I, _I, _P, _I, _P, _P, _P]),
    "cw_flac_info": (_I, [_P, C.c_int64, _P, _P, _P, _P]),
    "cw_flac_decode": (_I, [_P, C.c_int64, _P, C.c_int64, _P]),
    "cw_flac_last_error": (C.c_char_p, []),
    "cw_token_timestamps": (_I, [_P, _I, _I, _I, _P, _P]),
    "cw_beam_begin": (_I, [_P, _I, _I, _P, _I, _I, _I]),
    "cw_beam_step": (_I, [_P, _I, _P, _P]),
    "cw_beam_advance": (_I, [_P, _P, _P]),
    "cw_beam_finish": (_I, [_P, _I, _I, _P]),
    "cw_transcribe": (_I, [_P, _I, _P, C.POINTER(TranscribeCfg), _P, _P, _P, _I, _P]),
    "cw_align_matrix": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _P]),
    "cw_dtw": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "cw_adjust_pauses": (_I, [_P, _P, _P, _I, C.c_double]),
    "cw_vocab_create": (_P, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I]),
    "cw_vocab_destroy": (None, [_P]),
    "cw_collate_begin": (_P, [_P, C.c_double]),
    "cw_beam_host_new": (_P, [_I, _I, _I, _I, _I, _I, _I, C.c_double, _I, _P]),
    "cw_beam_host_step": (_I, [_P, _P, _P, _P, _P]),
    "cw_beam_host_result": (_I, [_P, _P, _P, _P]),
    "cw_beam_host_free": (None, [_P]),
    "cw_collate_set_mode": (_I, [_P, _I]),
    "cw_collate_feed": (_I, [_P, _P, _I, _P, _I, _I, C.c_double, C.c_double, C.c_double]),
    "cw_collate_finish": (_I, [_P, _P, _P, _P, _P]),
    "cw_collate_get": (_I, [_P, _P, _P, _P, _P, _P]),
    "cw_collate_free": (None, [_P]),
    "cw_set_option": (_I, [_P, C.c_char_p, _I]),
    "cw_test_set_option": (_I, [C.c_char_p, _I]),
    "cw_test_gemm": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _P]),
    "cw_test_gemm_fp8": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _P]),
    "cw_test_gemv": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P]),
    "cw_has_experiments": (_I, []),
    "cw_test_skinny": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P]),
    "cw_test_attention": (_I, [_P, _I, _I, _I, _P, _P, _P, _P]),
    "cw_test_cross_attention": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P]),
    "cw_test_sample": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "cw_stage_times": (_I, [_P, _P, _P, _I]),
    "cw_time_kernel": (_I, [_P, _I, _I, _I, _P, _P]),
    "cw_time_decode_stage": (_I, [_P, _I, _I, _I, _P, _P, _P, _P]),
    "cw_decode_stage_name": (C.c_char_p, [_I]),
    "cw_handoff_fallbacks": (_I, [_P]),
    "cw_handoff_resumes": (_I, [_P]),
    "cw_decode_stage_launches": (_I, [_P, _I]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGS)


def load():
    """Load the shared library and attach prototypes.  Raises NativeLibraryError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C crisperwhisper_amd/csrc` (hipcc, gfx950).  There is no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
