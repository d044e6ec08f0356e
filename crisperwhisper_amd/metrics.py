"""Timestamp-accuracy harness for BASELINE config 5 (word-timestamp F1 / mean IoU at a collar), to be run when
real CrisperWhisper weights and annotated DE/EN speech are available (neither is in this image).

Definition follows the reference's evaluation description (REF/README.md:78-90: "F1 Score / Avg IOU" with a collar):
reference and hypothesis word boundaries are matched one-to-one in time order; a hypothesis boundary is a hit if
it lies within ``collar`` seconds of an unmatched reference boundary.  IoU is averaged over words matched by text
alignment (same index after a longest-common-subsequence alignment of the word strings).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple


def _boundaries(words: Sequence[Dict]) -> List[float]:
    out: List[float] = []
    for w in words:
        out.extend([float(w["timestamp"][0]), float(w["timestamp"][1])])
    return sorted(out)


def boundary_f1(ref: Sequence[Dict], hyp: Sequence[Dict], collar: float = 0.2) -> Tuple[float, float, float]:
    """(precision, recall, f1) of hypothesis word boundaries against reference boundaries within +-collar."""
    r, h = _boundaries(ref), _boundaries(hyp)
    i = j = hits = 0
    while i < len(r) and j < len(h):
        if abs(r[i] - h[j]) <= collar + 1e-12:
            hits += 1; i += 1; j += 1
        elif h[j] < r[i]:
            j += 1
        else:
            i += 1
    p = hits / len(h) if h else 0.0
    rec = hits / len(r) if r else 0.0
    return p, rec, (2 * p * rec / (p + rec) if p + rec else 0.0)


def _lcs_pairs(a: List[str], b: List[str]) -> List[Tuple[int, int]]:
    n, m = len(a), len(b)
    dp = [[0] * (m + 1) for _ in range(n + 1)]
    for i in range(n - 1, -1, -1):
        for j in range(m - 1, -1, -1):
            dp[i][j] = dp[i + 1][j + 1] + 1 if a[i] == b[j] else max(dp[i + 1][j], dp[i][j + 1])
    i = j = 0
    pairs = []
    while i < n and j < m:
        if a[i] == b[j]:
            pairs.append((i, j)); i += 1; j += 1
        elif dp[i + 1][j] >= dp[i][j + 1]:
            i += 1
        else:
            j += 1
    return pairs


def mean_iou(ref: Sequence[Dict], hyp: Sequence[Dict]) -> float:
    """Mean temporal IoU over words aligned by text."""
    pairs = _lcs_pairs([w["text"].strip() for w in ref], [w["text"].strip() for w in hyp])
    if not pairs:
        return 0.0
    tot = 0.0
    for i, j in pairs:
        (a0, a1), (b0, b1) = ref[i]["timestamp"], hyp[j]["timestamp"]
        inter = max(0.0, min(a1, b1) - max(a0, b0))
        union = max(a1, b1) - min(a0, b0)
        tot += inter / union if union > 0 else (1.0 if inter == 0 and a0 == b0 else 0.0)
    return tot / len(pairs)
