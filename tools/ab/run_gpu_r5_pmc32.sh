#!/bin/bash
# round 5: PMC passes of the headline command at --tokens 32 (the largest count at which rocprofv3 survives a TCC counter here:
# it segfaults at 128 and hangs at 64, see run_gpu_r5_pmc.sh).  usage: run_gpu_r5_pmc32.sh TAG
TAG=${1:-r5p}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/pmc2; export PYTHONUNBUFFERED=1
ARGS="--batch 8 --tokens 32 --steps 1 --warmup 0 --no-cpu-baseline --no-longform --no-config3 --no-rccl --kernel-iters 3"
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc2 -o fetch -- python $R/bench.py $ARGS > $R/gpurun_out/pmc2_fetch.log 2>&1; echo "FETCH_SIZE rc=$?"
timeout 240 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc2 -o write -- python $R/bench.py $ARGS > $R/gpurun_out/pmc2_write.log 2>&1; echo "WRITE_SIZE rc=$?"
timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc2 -o mfma -- python $R/bench.py $ARGS > $R/gpurun_out/pmc2_mfma.log 2>&1; echo "MFMA rc=$?"
cd $R
python profiles/summarize_pmc.py gpurun_out/pmc2/fetch_results.db gpurun_out/pmc2/write_results.db gpurun_out/pmc2/mfma_results.db "python bench.py $ARGS" > gpurun_out/${TAG}_pmc_traffic_B8.json && head -c 400 gpurun_out/${TAG}_pmc_traffic_B8.json
rm -rf gpurun_out/pmc2
