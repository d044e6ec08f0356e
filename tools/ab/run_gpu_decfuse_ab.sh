#!/bin/bash
# fused decoder stages: targeted tests + kernel table + bench A/B against the eight-launch layer (CW_NO_FUSE6=1) and the
# rejected six-launch variant (CW_FUSE_MLP=1)
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "teacher_forced or bf16_runs or large_geometry_layers or full_size_bf16 or bench_shape or reproducible" 2>&1 | tail -8
for v in "base A=1" ${DECFUSE_AB_ALL:+"mlp CW_FUSE_MLP=1" "old CW_NO_FUSE6=1"}; do
  set -- $v
  bash tools/ab/run_gpu_prof2.sh dbg_$1 $2 -- > /dev/null 2>&1
  echo "== $v"; grep -E "cross_|gemv_stack|fc2x|gemv2_bf16|attn_decode" gpurun_out/prof_dbg_$1.txt | cut -c1-75,100-160
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_dbg_$1.log | head -1; grep -o '"parity": {[^}]*}' gpurun_out/prof_dbg_$1.log | grep -o '"clips_with[^}]*'
done
