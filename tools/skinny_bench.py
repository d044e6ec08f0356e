"""Microbenchmark of the 17..64-row decoder projections (csrc/skinny.hip; HIP events, weights rotating through copies larger than
the Infinity Cache): the six projections of a large-v3 decoder layer at 40 rows (8 items x 5 beams) and 64 rows (BASELINE
configs[3]), every K split the kernel supports.  Development aid; output -> profiles/r04_skinny_sweep.txt.

    python tools/skinny_bench.py [rows ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crisperwhisper_amd import synthetic as syn
from crisperwhisper_amd.engine import Engine
from tests import helpers as Hh

SHAPES = [("qkv   (LN, planes)", 0, 3840, 1280), ("q_c   (LN, planes)", 0, 1280, 1280), ("fc1   (LN, planes)", 1, 5120, 1280),
          ("o/c-o (atomics)   ", 2, 1280, 1280), ("fc2   (atomics)   ", 2, 1280, 5120)]


def main():
    rows = [int(a) for a in sys.argv[1:]] or [40, 64]
    g, v, W, spec = Hh.tiny_setup()
    eng = Engine(spec, dtype="bf16", max_batch=4)
    rng = np.random.default_rng(0)
    try:
        for Mb in rows:
            print(f"# rows = {Mb}")
            for name, mode, N, K in SHAPES:
                x = rng.standard_normal((Mb, K)).astype(np.float32)
                Wm = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
                for nks in (1, 2, 4, 5, 8, 10, 16):
                    if (K // 32) % nks:
                        continue
                    S = (K // 32) // nks
                    if mode != 2 and S > 20:
                        continue
                    best = None
                    for _ in range(2):
                        _, us = eng.test_skinny(mode, x, Wm, None, nks=nks, reps=200)
                        best = us if best is None or us[0] + us[1] < best[0] + best[1] else best
                    blocks = ((N + 63) // 64) * S
                    kb_w = 64 * nks * 32 * 2 / 1024
                    kb_a = ((Mb + 15) // 16) * 16 * nks * 32 * (2 if mode == 2 else 4) / 1024
                    print(f"{name} N={N:5d} K={K:5d} nks={nks:2d} S={S:3d} blocks={blocks:4d} W/blk={kb_w:5.0f}KB act/blk={kb_a:5.0f}KB  "
                          f"gemm {best[0]:6.2f} us  finish {best[1]:5.2f} us  total {best[0] + best[1]:6.2f} us"
                          f"  weights at {N * K * 2 / best[0] / 1e6:5.2f} TB/s", flush=True)
    finally:
        eng.close()


if __name__ == "__main__":
    main()
