"""Chunk-level data parallelism: independent 30 s chunks are sharded in contiguous blocks across the
ranks of one node (one process per GPU); the only communication is one small all-gather of fixed-size
per-chunk records (token ids + token timestamps + stride) over torch.distributed (RCCL on GPUs, gloo
in CPU tests); every rank then holds all records in audio order."""
from __future__ import annotations

from typing import List, Optional

import numpy as np

REC_TOKENS = 448
REC_WORDS = 6 + 2 * REC_TOKENS   # chunk_idx, n_tok, n_ts, stride(3 x f32 bits), tokens, ts bits


def shard_bounds(n_chunks: int, world: int):
    """Contiguous blocks, sizes differ by at most one (30 chunks / 8 ranks -> 4,4,4,4,4,4,3,3)."""
    base, rem = divmod(n_chunks, world)
    bounds, s = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        bounds.append((s, s + n))
        s += n
    return bounds


def pack_record(idx: int, tokens: np.ndarray, ts: np.ndarray, stride) -> np.ndarray:
    rec = np.zeros(REC_WORDS, dtype=np.int32)
    nt, ns = len(tokens), len(ts)
    if nt > REC_TOKENS or ns > REC_TOKENS:
        raise ValueError("chunk output exceeds max_target_positions")
    rec[0], rec[1], rec[2] = idx, nt, ns
    rec[3:6] = np.asarray(stride, dtype=np.float32).view(np.int32)
    rec[6:6 + nt] = tokens
    rec[6 + REC_TOKENS:6 + REC_TOKENS + ns] = np.asarray(ts, dtype=np.float32).view(np.int32)
    return rec


def unpack_record(rec: np.ndarray):
    idx, nt, ns = int(rec[0]), int(rec[1]), int(rec[2])
    stride = tuple(float(x) for x in rec[3:6].view(np.float32))
    tokens = rec[6:6 + nt].astype(np.int64)
    ts = rec[6 + REC_TOKENS:6 + REC_TOKENS + ns].view(np.float32).copy()
    return idx, tokens, ts, stride


class Shard:
    """rank/world + the gather primitive."""

    def __init__(self, rank: int = 0, world: int = 1, device: Optional[str] = None):
        self.rank, self.world, self.device = rank, world, device

    def all_gather_records(self, recs: np.ndarray, max_per_rank: int) -> np.ndarray:
        """recs [n_local, REC_WORDS] int32 -> [n_total, REC_WORDS] ordered by chunk index."""
        if self.world == 1:
            return recs
        import torch
        import torch.distributed as dist
        buf = np.full((max_per_rank, REC_WORDS), -1, dtype=np.int32)
        buf[:len(recs)] = recs
        t = torch.from_numpy(buf)
        if self.device is not None:
            t = t.to(self.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        allr = torch.stack(out).cpu().numpy().reshape(-1, REC_WORDS)
        allr = allr[allr[:, 0] >= 0]
        return allr[np.argsort(allr[:, 0], kind="stable")]
