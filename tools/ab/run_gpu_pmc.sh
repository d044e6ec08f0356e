#!/bin/bash
# PMC passes of the bench command (FETCH_SIZE / WRITE_SIZE / MFMA busy: three separate rocprofv3 runs, counters only).
# usage: run_gpu_pmc.sh OUT.json [extra bench args]
OUT=${1:-gpurun_out/pmc_traffic.json}; shift
mkdir -p gpurun_out/pmc2
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
ARGS="--batch 8 --tokens 4 --steps 1 --warmup 0 --no-cpu-baseline --no-longform --no-config3 --kernel-iters 3 $*"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc2 -o fetch -- python $R/bench.py $ARGS > $R/gpurun_out/pmc2_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc2 -o write -- python $R/bench.py $ARGS > $R/gpurun_out/pmc2_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc2 -o mfma -- python $R/bench.py $ARGS > $R/gpurun_out/pmc2_mfma.log 2>&1
cd $R
python profiles/summarize_pmc.py gpurun_out/pmc2/fetch_results.db gpurun_out/pmc2/write_results.db gpurun_out/pmc2/mfma_results.db "python bench.py $ARGS" > $OUT && head -c 2500 $OUT
rm -rf gpurun_out/pmc2
