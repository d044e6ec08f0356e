"""Bit-level FLAC *writer* used only by the tests of the native FLAC decoder (crisperwhisper_amd/csrc/flac.cpp).

Written independently from the format specification (RFC 9639) so that decoder and test vector generator do not share code:
numpy / Python integers here, a streaming C++ bit reader there.  Every construct the decoder implements can be emitted on
request: CONSTANT / VERBATIM / FIXED (order 0-4) / LPC (order 1-32) subframes, Rice and Rice2 partitions of any order incl.
escape partitions, wasted bits, independent / left-side / side-right / mid-side stereo, 8..32 bits per sample, every block-size
and sample-rate header code, multi-byte coded frame numbers, extra metadata blocks, and the STREAMINFO MD5 signature."""
from __future__ import annotations

import hashlib
from typing import List, Optional, Sequence

import numpy as np


class BitWriter:
    def __init__(self):
        self.bits: List[int] = []          # list of (value, nbits)
        self.acc = 0
        self.n = 0
        self.out = bytearray()

    def u(self, v: int, nbits: int):
        if nbits == 0:
            return
        assert 0 <= v < (1 << nbits), (v, nbits)
        self.acc = (self.acc << nbits) | v
        self.n += nbits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def s(self, v: int, nbits: int):
        assert -(1 << (nbits - 1)) <= v < (1 << (nbits - 1)), (v, nbits)
        self.u(v & ((1 << nbits) - 1), nbits)

    def unary(self, q: int):
        while q >= 32:
            self.u(0, 32); q -= 32
        self.u(1, q + 1)

    def align(self):
        if self.n:
            self.u(0, 8 - self.n)

    def bytes(self) -> bytes:
        assert self.n == 0
        return bytes(self.out)


def crc8(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def utf8_number(v: int) -> bytes:
    if v < 0x80:
        return bytes([v])
    n = 2
    while v >= (1 << (5 * n + 1)) and n < 7:      # 2 bytes: 11 bits, 3: 16, 4: 21, 5: 26, 6: 31, 7: 36
        n += 1
    out = []
    for _ in range(n - 1):
        out.append(0x80 | (v & 0x3F)); v >>= 6
    lead = ((0xFF << (8 - n)) & 0xFF) | v
    return bytes([lead] + out[::-1])


def _zigzag(r: int) -> int:
    return (r << 1) if r >= 0 else ((-r << 1) - 1)


def write_residual(bw: BitWriter, res: Sequence[int], n: int, order: int, method: int, porder: int, escape_parts=()):
    bw.u(method, 2)
    bw.u(porder, 4)
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    parts = 1 << porder
    idx = 0
    for p in range(parts):
        cnt = (n >> porder) - (order if p == 0 else 0) if porder else n - order
        chunk = [int(x) for x in res[idx: idx + cnt]]
        idx += cnt
        if p in escape_parts or not chunk:
            bw.u(esc, pbits)
            raw = max([1] + [(abs(x) if x >= 0 else abs(x + 1)).bit_length() + 1 for x in chunk]) if chunk else 0
            bw.u(raw, 5)
            for x in chunk:
                bw.s(x, raw)
            continue
        mean = sum(_zigzag(x) for x in chunk) / len(chunk)
        k = max(0, min(esc - 1, int(np.log2(mean + 1))))
        bw.u(k, pbits)
        for x in chunk:
            z = _zigzag(x)
            bw.unary(z >> k)
            bw.u(z & ((1 << k) - 1), k)
    assert idx == len(res)


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def write_subframe(bw: BitWriter, x: Sequence[int], bps: int, kind: str, *, order: int = 0, method: int = 0, porder: int = 0,
                   escape_parts=(), wasted: int = 0, lpc=None):
    x = [int(v) for v in x]
    n = len(x)
    if wasted:
        assert all(v % (1 << wasted) == 0 for v in x)
        x = [v >> wasted for v in x]
        bps -= wasted
    bw.u(0, 1)
    code = {"constant": 0, "verbatim": 1}.get(kind)
    if kind == "fixed":
        code = 8 + order
    elif kind == "lpc":
        code = 32 + order - 1
    bw.u(code, 6)
    if wasted:
        bw.u(1, 1); bw.unary(wasted - 1)
    else:
        bw.u(0, 1)
    if kind == "constant":
        assert all(v == x[0] for v in x)
        bw.s(x[0], bps)
    elif kind == "verbatim":
        for v in x:
            bw.s(v, bps)
    elif kind == "fixed":
        for v in x[:order]:
            bw.s(v, bps)
        co = FIXED[order]
        res = [x[i] - sum(c * x[i - 1 - k] for k, c in enumerate(co)) for i in range(order, n)]
        write_residual(bw, res, n, order, method, porder, escape_parts)
    elif kind == "lpc":
        coef, prec, shift = lpc
        for v in x[:order]:
            bw.s(v, bps)
        bw.u(prec - 1, 4)
        bw.s(shift, 5)
        for c in coef:
            bw.s(int(c), prec)
        res = [x[i] - (sum(int(c) * x[i - 1 - k] for k, c in enumerate(coef)) >> shift) for i in range(order, n)]
        write_residual(bw, res, n, order, method, porder, escape_parts)
    else:
        raise ValueError(kind)


BS_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
SR_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
SS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def write_frame(chans: List[np.ndarray], bps: int, sr: int, number: int, stereo: str, plans: List[dict], *, variable=False,
                use_streaminfo_bps=False, use_streaminfo_sr=False) -> bytes:
    n = len(chans[0])
    bw = BitWriter()
    bw.u(0x3FFE, 14); bw.u(0, 1); bw.u(1 if variable else 0, 1)
    bs_code = BS_CODES.get(n)
    bs_extra = None
    if bs_code is None:
        bs_code, bs_extra = (6, (n - 1, 8)) if n <= 256 else (7, (n - 1, 16))
    sr_code, sr_extra = SR_CODES.get(sr), None
    if use_streaminfo_sr:
        sr_code = 0
    elif sr_code is None:
        if sr % 1000 == 0 and sr // 1000 < 256:
            sr_code, sr_extra = 12, (sr // 1000, 8)
        elif sr < 65536:
            sr_code, sr_extra = 13, (sr, 16)
        else:
            sr_code, sr_extra = 14, (sr // 10, 16)
    bw.u(bs_code, 4); bw.u(sr_code, 4)
    ch_code = {"independent": len(chans) - 1, "left_side": 8, "side_right": 9, "mid_side": 10}[stereo]
    bw.u(ch_code, 4)
    bw.u(0 if use_streaminfo_bps else SS_CODES[bps], 3); bw.u(0, 1)
    for b in utf8_number(number):
        bw.u(b, 8)
    if bs_extra:
        bw.u(*bs_extra)
    if sr_extra:
        bw.u(*sr_extra)
    hdr = bw.bytes()
    bw.u(crc8(hdr), 8)
    cs = [c.astype(object) for c in chans]
    widths = [bps] * len(cs)
    if stereo == "left_side":
        cs = [cs[0], cs[0] - cs[1]]; widths = [bps, bps + 1]
    elif stereo == "side_right":
        cs = [cs[0] - cs[1], cs[1]]; widths = [bps + 1, bps]
    elif stereo == "mid_side":
        side = cs[0] - cs[1]
        mid = np.array([(int(a) + int(b)) >> 1 for a, b in zip(cs[0], cs[1])], dtype=object)
        cs = [mid, side]; widths = [bps, bps + 1]
    for c, w, plan in zip(cs, widths, plans):
        write_subframe(bw, list(c), w, **plan)
    bw.align()
    body = bw.bytes()
    return body + crc16(body).to_bytes(2, "big")


def streaminfo(min_bs, max_bs, sr, nch, bps, total, md5: bytes) -> bytes:
    bw = BitWriter()
    bw.u(min_bs, 16); bw.u(max_bs, 16); bw.u(0, 24); bw.u(0, 24)
    bw.u(sr, 20); bw.u(nch - 1, 3); bw.u(bps - 1, 5); bw.u(total, 36)
    return bw.bytes() + md5


def md5_of(chans: List[np.ndarray], bps: int) -> bytes:
    nb = (bps + 7) // 8
    inter = np.stack([c.astype(np.int64) for c in chans], axis=1).reshape(-1)
    raw = b"".join(int(v).to_bytes(nb, "little", signed=True) for v in inter)
    return hashlib.md5(raw).digest()


def write_stream(chans: List[np.ndarray], bps: int, sr: int, frames: List[dict], *, with_md5=True, total_known=True,
                 extra_blocks: Sequence[tuple] = (), first_number: int = 0) -> bytes:
    """frames: [{"n": blocksize, "stereo": ..., "plans": [subframe plan per channel], optional header switches}]."""
    total = len(chans[0])
    assert sum(f["n"] for f in frames) == total
    md5 = md5_of(chans, bps) if with_md5 else bytes(16)
    sizes = [f["n"] for f in frames]
    blocks = [(0, streaminfo(min(sizes), max(sizes), sr, len(chans), bps, total if total_known else 0, md5))] + list(extra_blocks)
    out = bytearray(b"fLaC")
    for i, (t, body) in enumerate(blocks):
        out += bytes([(0x80 if i == len(blocks) - 1 else 0) | t]) + len(body).to_bytes(3, "big") + body
    pos = 0
    variable = any(f.get("variable") for f in frames)
    for k, f in enumerate(frames):
        seg = [c[pos: pos + f["n"]] for c in chans]
        number = pos if variable else first_number + k
        out += write_frame(seg, bps, sr, number, f.get("stereo", "independent"), f["plans"], variable=variable,
                           use_streaminfo_bps=f.get("use_streaminfo_bps", False), use_streaminfo_sr=f.get("use_streaminfo_sr", False))
        pos += f["n"]
    return bytes(out)


def lpc_plan(x: np.ndarray, order: int, prec: int = 12):
    """Quantised LPC coefficients from the autocorrelation method (what an encoder does; any coefficients would be valid)."""
    xf = x.astype(np.float64)
    r = np.array([np.dot(xf[:len(xf) - k], xf[k:]) for k in range(order + 1)])
    r[0] = r[0] * (1 + 1e-9) + 1e-9
    R = np.array([[r[abs(i - j)] for j in range(order)] for i in range(order)])
    a = np.linalg.solve(R + 1e-6 * r[0] * np.eye(order), r[1:order + 1])
    amax = max(np.abs(a).max(), 1e-9)
    shift = int(max(0, min(15, prec - 2 - int(np.ceil(np.log2(amax + 1e-12))))))
    q = np.clip(np.round(a * (1 << shift)), -(1 << (prec - 1)), (1 << (prec - 1)) - 1).astype(np.int64)
    return [int(v) for v in q], prec, shift
