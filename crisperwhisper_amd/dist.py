"""Chunk-level data parallelism: independent 30 s chunks are sharded in contiguous blocks across the
ranks of one node (one process per GPU); the only communication is one small all-gather of fixed-size
per-chunk records (token ids + token timestamps + stride) over torch.distributed (RCCL on GPUs, gloo
in CPU tests); every rank then holds all records in audio order."""
from __future__ import annotations

from typing import List, Optional

import numpy as np

REC_TOKENS = 4 * 448        # a 30 s chunk is usually one generate pass (<= 445 tokens); the seek loop can add more
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
        raise ValueError(f"chunk output of {max(nt, ns)} tokens exceeds the record capacity ({REC_TOKENS})")
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


WORD_MAX = 448
WORD_TEXT_BYTES = 4096
WORD_REC = 3 + 4 * WORD_MAX + WORD_TEXT_BYTES // 4   # chunk_idx, n_words, n_text_bytes, (start,end) f64 bits, utf-8 text


def pack_words(idx: int, words) -> np.ndarray:
    """Per-chunk *word list* record (what the ranks exchange when chunks are independent clips):
    words = [{"text", "timestamp": (start, end)}].  Texts are joined with a 0x00 separator."""
    rec = np.zeros(WORD_REC, dtype=np.int32)
    if len(words) > WORD_MAX:
        raise ValueError("too many words in one chunk")
    blob = b"\x00".join(w["text"].encode("utf-8") for w in words)
    if len(blob) > WORD_TEXT_BYTES:
        raise ValueError("word text exceeds record capacity")
    rec[0], rec[1], rec[2] = idx, len(words), len(blob)
    ts = np.asarray([w["timestamp"] for w in words], dtype=np.float64).reshape(-1)
    rec[3:3 + 2 * len(ts)] = ts.view(np.int32)
    buf = np.zeros(WORD_TEXT_BYTES, dtype=np.uint8)
    buf[:len(blob)] = np.frombuffer(blob, dtype=np.uint8)
    rec[3 + 4 * WORD_MAX:] = buf.view(np.int32)
    return rec


def unpack_words(rec: np.ndarray):
    idx, n, nb = int(rec[0]), int(rec[1]), int(rec[2])
    ts = rec[3:3 + 4 * n].view(np.float64).reshape(n, 2)
    blob = rec[3 + 4 * WORD_MAX:].view(np.uint8)[:nb].tobytes()
    texts = blob.decode("utf-8").split("\x00") if n else []
    return idx, [{"text": t, "timestamp": (float(a), float(b))} for t, (a, b) in zip(texts, ts)]


class Shard:
    """rank/world + the gather primitive."""

    def __init__(self, rank: int = 0, world: int = 1, device: Optional[str] = None, collective_at_world1: bool = False):
        """``collective_at_world1``: run the all-gather through the process group even when it has a single rank (the
        1-GPU bench then exercises the RCCL path end to end instead of short-cutting it)."""
        self.rank, self.world, self.device = rank, world, device
        self.collective_at_world1 = collective_at_world1
        self.n_collectives = 0
        self.gather_s = 0.0            # host wall time spent inside all_gather_records (staging copy + collective + copy back)

    def all_gather_records(self, recs: np.ndarray, max_per_rank: int) -> np.ndarray:
        """recs [n_local, W] int32 (W = REC_WORDS or WORD_REC) -> [n_total, W] ordered by chunk index."""
        if self.world == 1 and not self.collective_at_world1:
            return recs
        self.n_collectives += 1
        import time
        import torch
        import torch.distributed as dist
        t_g0 = time.perf_counter()
        REC_WORDS = recs.shape[1]
        buf = np.full((max_per_rank, REC_WORDS), -1, dtype=np.int32)
        buf[:len(recs)] = recs
        t = torch.from_numpy(buf)
        if self.device is not None:
            t = t.to(self.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        allr = torch.stack(out).cpu().numpy().reshape(-1, REC_WORDS)
        self.gather_s += time.perf_counter() - t_g0
        allr = allr[allr[:, 0] >= 0]
        return allr[np.argsort(allr[:, 0], kind="stable")]
