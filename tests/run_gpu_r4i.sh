#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --timeout=600 -p no:cacheprovider -k "second_weight_seed or fused_decoder_stage" 2>&1 | grep -E "AssertionError|passed|failed|^E  " | head -20
