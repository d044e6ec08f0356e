#!/bin/bash
# round 4: tile order of the 8-phase GEMM against the L2 fetch counter (the "6.4 x operands" question of the round-3 review).
# Needs the experiments build (option "gemm_gm").  Settings: g > 0 = m-tiles per group with the XCD remap, g < 0 = groups of -g
# without the remap.  Timing from cw_test_gemm's event loop, FETCH_SIZE from one rocprofv3 counter pass (one dispatch each).
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cat > /tmp/tile_ab.py <<'P'
import os, sys, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import helpers as Hh
from crisperwhisper_amd.engine import Engine
g, v, W, spec = Hh.tiny_setup()
e = Engine(spec, dtype="bf16", max_batch=4)
rng = np.random.default_rng(0)
shapes = [(12000, 5120, 1280, True), (12000, 1280, 5120, False), (12000, 3840, 1280, False)]
mats = [(rng.standard_normal((M, K)).astype(np.float32), (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32), gelu) for (M, N, K, gelu) in shapes]
ref = None
for gm in (8, -8, 2, 4, 16, 47):
    assert e.lib.cw_test_set_option(b"gemm_gm", gm) == 0
    sys.stderr.write(f"== gm {gm}\n")
    outs = [e.test_gemm(A, Wm, None, gelu) for (A, Wm, gelu) in mats]
    if ref is None: ref = outs
    else: assert all(np.array_equal(a, b) for a, b in zip(ref, outs)), "tile order changed a result"
e.close()
P
[ -n "$SKIP_TIMING" ] || CW_TEST_GEMM_REPS=20 timeout 600 python /tmp/tile_ab.py 2>&1 | grep "==\|cw_test_gemm\|Error\|rror" | tee gpurun_out/r4_tile_order_time.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_tile
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_tile -o run -- python /tmp/tile_ab.py > $R/gpurun_out/pmc_tile.log 2>&1
cd $R
python tools/pmc_table.py gpurun_out/pmc_tile/run_results.db 8ph --each > gpurun_out/r4_tile_order_fetch.txt 2>&1
cat gpurun_out/r4_tile_order_fetch.txt | head -40
rm -rf gpurun_out/pmc_tile
