#!/bin/bash
# SQ counters for the encoder attention kernel at the bench shape (counters only, one group per rocprofv3 pass).
TAG=${1:-pmc}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/attn_pmc
cat > /tmp/attn_one.py <<'P'
import os, sys, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"] + "/tests")
import helpers as Hh
from crisperwhisper_amd.engine import Engine
g, v, W, spec = Hh.tiny_setup()
e = Engine(spec, dtype="bf16", max_batch=4)
rng = np.random.default_rng(0)
B, H, S = 8, 20, 1500
q = (rng.standard_normal((B, H, S, 64)) * 0.3).astype(np.float32)
k = rng.standard_normal((B, H, S, 64)).astype(np.float32)
vv = rng.standard_normal((B, H, S, 64)).astype(np.float32)
e.test_attention(q, k, vv)
e.close()
P
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  CW_TEST_ATTN_REPS=3 timeout 300 rocprofv3 --pmc $grp -d $R/gpurun_out/attn_pmc -o g$i --output-format csv -- python /tmp/attn_one.py > $R/gpurun_out/attn_pmc/g$i.log 2>&1
done
cd $R
python - <<'P' | tee gpurun_out/attn_pmc_$TAG.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/attn_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_encoder" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for n, v in sorted(d.items()):
        print(f"   {n:36s} n={len(v):3d} avg={sum(v)/len(v):16.1f}")
P
rm -rf gpurun_out/attn_pmc
