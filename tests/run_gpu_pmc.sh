#!/bin/bash
mkdir -p gpurun_out/pmc
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc -o fetch -- python $R/bench.py --batch 8 --tokens 4 --steps 1 --warmup 0 --no-cpu-baseline --kernel-iters 3 > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc -o write -- python $R/bench.py --batch 8 --tokens 4 --steps 1 --warmup 0 --no-cpu-baseline --kernel-iters 3 > $R/gpurun_out/pmc_write.log 2>&1
ls -la $R/gpurun_out/pmc
# MFMA pipe occupancy of the encoder kernels (north_star: "MFMA utilisation for the transformer"): busy cycles of the MFMA
# pipe against GPU-active cycles, one more pass
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc -o mfma -- python $R/bench.py --batch 8 --tokens 4 --steps 1 --warmup 0 --no-cpu-baseline --kernel-iters 3 > $R/gpurun_out/pmc_mfma.log 2>&1
ls -la $R/gpurun_out/pmc
