#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python bench.py "$@" > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log | cut -c1-3500
