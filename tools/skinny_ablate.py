"""Where a skinny-M projection launch spends its time (csrc/skinny.hip built with `make EXTRA=-DCW_SK_DEBUG`): the same launch with
the activation staging (1), the MFMA loop (2), the epilogue stores (4) or the weight stream (8) switched off, 64 rows."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crisperwhisper_amd.engine import Engine
from tests import helpers as Hh

CASES = [("qkv  planes", 0, 3840, 1280, 10), ("fc1  planes", 1, 5120, 1280, 8), ("o    atomic", 2, 1280, 1280, 4), ("fc2  atomic", 2, 1280, 5120, 16)]
g, v, W, spec = Hh.tiny_setup()
eng = Engine(spec, dtype="bf16", max_batch=4)
rng = np.random.default_rng(0)
Mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, mode, N, K, nks in CASES:
    x = rng.standard_normal((Mb, K)).astype(np.float32)
    Wm = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    line = f"{name} N={N} K={K} nks={nks}: "
    for dbg in (0, 1, 2, 4, 8, 3, 7, 15):
        os.environ["CW_SK_DBG"] = str(dbg)
        best = min(eng.test_skinny(mode, x, Wm, None, nks=nks, reps=300)[1][0] for _ in range(2))
        line += f" dbg{dbg}={best:5.2f}"
    _, us = eng.test_skinny(2, x[:, :128].copy(), Wm[:64, :128].copy(), None, nks=4, reps=300)
    print(line + f"   empty launch {-us[1]:.2f} us", flush=True)
eng.close()
