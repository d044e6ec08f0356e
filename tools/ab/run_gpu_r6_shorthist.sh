export CW_LIB_PATH=$PWD/crisperwhisper_amd/libcw_new.so
python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "batch64 or odd_batch or beam or reproducible or large_batch or reference_batch_size_16 or bench_shape" 2>&1 | tail -3
B="--steps 2 --warmup 1 --no-cpu-baseline --no-longform --no-config3"
for i in 1 2; do for e in X=1 CW_NO_SHORT_HIST=1; do
 env $e python bench.py $B --batch 64 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('b64 $e', round(l['ms_per_step'],1), round(l['stage_roofline']['decode_step']['ms_per_step'],4), l['parity']['clips_with_identical_text'])"
 env $e python bench.py $B --batch 16 --steps 3 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('b16 $e', round(l['ms_per_step'],1), round(l['stage_roofline']['decode_step']['ms_per_step'],4), l['parity']['clips_with_identical_text'])"
 env $e python tools/beam_step_bench.py | tail -1 | cut -c60-140
done; done
