#!/bin/bash
# round 6 (as round 5, starting at the token count that is known to survive): PMC passes of the headline command (FETCH_SIZE / WRITE_SIZE / MFMA busy: three separate counters-only rocprofv3 runs) at
# the largest token count rocprofv3 survives: with a TCC counter it segfaults a few seconds into --tokens 128 (27 k dispatches per
# pass; the SQ / GRBM pass is fine), so the token count is halved until the FETCH_SIZE pass completes.
# usage: run_gpu_r6_pmc.sh TAG
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/pmc2; export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
for T in 32 16 8; do
  ARGS="--batch 8 --tokens $T --steps 1 --warmup 0 --no-cpu-baseline --no-longform --no-config3 --no-rccl --kernel-iters 3"
  rm -f $R/gpurun_out/pmc2/fetch*
  timeout 900 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc2 -o fetch -- python $R/bench.py $ARGS > $R/gpurun_out/pmc2_fetch.log 2>&1
  rc=$?; echo "FETCH_SIZE pass at --tokens $T: rc=$rc"
  [ $rc -eq 0 ] && break
done
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc2 -o write -- python $R/bench.py $ARGS > $R/gpurun_out/pmc2_write.log 2>&1
echo "WRITE_SIZE rc=$?"
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc2 -o mfma -- python $R/bench.py $ARGS > $R/gpurun_out/pmc2_mfma.log 2>&1
cd $R
python profiles/summarize_pmc.py gpurun_out/pmc2/fetch_results.db gpurun_out/pmc2/write_results.db gpurun_out/pmc2/mfma_results.db "python bench.py $ARGS" > gpurun_out/${TAG}_pmc_traffic_B8.json && head -c 600 gpurun_out/${TAG}_pmc_traffic_B8.json
rm -rf gpurun_out/pmc2
