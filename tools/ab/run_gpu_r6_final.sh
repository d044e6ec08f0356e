#!/bin/bash
# end-of-round measurements: the driver's bench command, kernel tables of the three shapes (B = 8, B = 64, beam 8 x 5), B = 16
mkdir -p gpurun_out
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r06_final_bench_default.json 2> gpurun_out/r06_final_bench_default.err ) 2>&1 | tail -3
HEADN=30 bash tools/ab/run_gpu_prof_args.sh r6final_B8 --no-config3 > /dev/null
HEADN=30 bash tools/ab/run_gpu_prof_args.sh r6final_B64 --no-config3 --batch 64 > /dev/null
HEADN=30 bash tools/ab/run_gpu_prof_args.sh r6final_B16 --no-config3 --batch 16 > /dev/null
HEADN=30 bash tools/ab/run_gpu_prof_cmd.sh r6final_beam python tools/beam_step_bench.py --reps 2 > /dev/null
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --batch 16 > gpurun_out/r06_final_bench_B16.json 2>/dev/null
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-longform --no-config3 --num-beams 5 > gpurun_out/r06_final_bench_beam5.json 2>/dev/null
python tools/beam_step_bench.py 2>&1 | tail -1 > gpurun_out/r06_final_beam_step.txt
ls gpurun_out | grep r6final
