"""Token-timestamp extraction: z-score -> median filter -> head mean -> DTW -> jump times.
numpy restatement (test oracle) of TF/models/whisper/generation_whisper.py:43-61
(_median_filter), :64-115 (_dynamic_time_warping), :241-381 (_extract_token_timestamps).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _dtw_lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "_build", "liboracle_dtw.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _LIB = ctypes.CDLL(so)
        _LIB.cw_oracle_dtw.restype = ctypes.c_int
    return _LIB


def median_filter(x: np.ndarray, width: int) -> np.ndarray:
    """:43-61 -- reflect pad width//2, sliding window, sorted middle element; early return when
    the last dim is <= width//2."""
    if width <= 0 or width % 2 != 1:
        raise ValueError("`filter_width` should be an odd number")
    pad = width // 2
    if x.shape[-1] <= pad:
        return x
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(pad, pad)], mode="reflect")
    win = np.lib.stride_tricks.sliding_window_view(xp, width, axis=-1)
    return np.sort(win, axis=-1)[..., pad]


def dtw_python(matrix: np.ndarray):
    """Line-by-line restatement of :64-115 (slow; small cases only)."""
    N, M = matrix.shape
    cost = np.ones((N + 1, M + 1), dtype=np.float32) * np.inf
    trace = -np.ones((N + 1, M + 1), dtype=np.float32)
    cost[0, 0] = 0
    for j in range(1, M + 1):
        for i in range(1, N + 1):
            c0, c1, c2 = cost[i - 1, j - 1], cost[i - 1, j], cost[i, j - 1]
            if c0 < c1 and c0 < c2:
                c, t = c0, 0
            elif c1 < c0 and c1 < c2:
                c, t = c1, 1
            else:
                c, t = c2, 2
            cost[i, j] = matrix[i - 1, j - 1] + c
            trace[i, j] = t
    i, j = N, M
    trace[0, :] = 2
    trace[:, 0] = 1
    ti, tj = [], []
    while i > 0 or j > 0:
        ti.append(i - 1)
        tj.append(j - 1)
        if trace[i, j] == 0:
            i -= 1; j -= 1
        elif trace[i, j] == 1:
            i -= 1
        elif trace[i, j] == 2:
            j -= 1
        else:
            raise RuntimeError("Internal error in dynamic time warping.")
    return np.array(ti)[::-1], np.array(tj)[::-1]


def dtw(matrix: np.ndarray):
    """C-accelerated, same arithmetic as dtw_python (oracle/dtw.c)."""
    m = np.ascontiguousarray(matrix, dtype=np.float64)
    N, M = m.shape
    ti = np.empty(N + M + 2, dtype=np.int32)
    tj = np.empty(N + M + 2, dtype=np.int32)
    n = _dtw_lib().cw_oracle_dtw(m.ctypes.data_as(ctypes.c_void_p), N, M,
                                 ti.ctypes.data_as(ctypes.c_void_p), tj.ctypes.data_as(ctypes.c_void_p))
    if n < 0:
        raise RuntimeError("Internal error in dynamic time warping.")
    return ti[:n].astype(np.int64), tj[:n].astype(np.int64)


def normalise_filter_mean(w: np.ndarray, median_width: int) -> np.ndarray:
    """[H, N, M] -> [N, M]: population z-score over tokens (:343-345), median over frames (:346),
    mean over heads (:349)."""
    w = w.astype(np.float32)
    std = w.std(axis=-2, keepdims=True, dtype=np.float32)        # unbiased=False
    mean = w.mean(axis=-2, keepdims=True, dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        z = ((w - mean) / std).astype(np.float32)
    z = median_filter(z, median_width)
    return z.mean(axis=0, dtype=np.float32)


def effective_cols(num_frames, S: int):
    """Columns of the [.., S] attention map that reach the DTW, with HF's exact slicing semantics:
    when every item has the same ``num_frames`` the map is cropped to ``[..., :nf // 2]`` for the whole
    batch (:318-323) *and again* per item inside the loop (:354); otherwise only per item.  With a
    positive ``nf // 2 <= S`` both give nf // 2; for ``nf <= 0`` (seek ran past the audio: a random
    model can emit a 29 s timestamp in a 12 s clip) Python's negative-stop slicing applies, possibly
    twice, and may leave zero columns."""
    nf = np.asarray(num_frames, dtype=np.int64)
    half = nf // 2                                   # floor division, like torch / Python
    uniform = len(np.unique(nf)) == 1
    out = []
    for h in half.tolist():
        n1 = len(range(S)[:h])
        out.append(len(range(n1)[:h]) if uniform else n1)
    return out


def extract_token_timestamps(weights: np.ndarray, num_frames, num_input_ids: int, median_width: int,
                             time_precision: float = 0.02) -> np.ndarray:
    """weights [B, H_a, L, S] f32 (alignment-head rows, prompt rows included) -> [B, L+1] f32.

    num_frames: per-item mel-frame counts (array) or None.  Mirrors :304-381 for the greedy
    (no beam_indices) path, including the double crop of the uniform case (see effective_cols)."""
    B, H, L, S = weights.shape
    ts = np.zeros((B, L + 1), dtype=np.float32)
    w = weights[:, :, num_input_ids:, :]
    N = w.shape[2]
    if N == 0:
        return ts
    cols = [S] * B if num_frames is None else effective_cols(num_frames, S)
    for b in range(B):
        nf = cols[b]
        if nf == 0:
            # _dynamic_time_warping on an [N, 0] matrix: backtrace walks up column 0, time index -1
            jump_times = np.full(N, -1 * time_precision)
        else:
            mat = normalise_filter_mean(w[b, :, :, :nf], median_width)
            ti, tj = dtw(-mat.astype(np.float64))
            jumps = np.pad(np.diff(ti), (1, 0), constant_values=1).astype(bool)
            jump_times = (tj[jumps] * time_precision)
        ts[b] = np.concatenate([np.zeros(num_input_ids), jump_times, jump_times[-1:]]).astype(np.float32)
    return ts
