#!/bin/bash
# same-box A/B of two builds of the library: the in-tree one ("old") against crisperwhisper_amd/libcw_new.so (make BUILD=build_new OUT=../libcw_new.so);
# usage: run_gpu_r6_libab.sh [extra bench args]   (three alternations)
B="--steps 5 --warmup 2 --no-cpu-baseline --no-longform --no-config3 $@"
for i in 1 2 3; do
for v in old new; do
  E="X=1"; [ $v = new ] && E="CW_LIB_PATH=$PWD/crisperwhisper_amd/libcw_new.so"
  env $E python bench.py $B 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('$v', round(l['ms_per_step'],2), round(l['stage_roofline']['decode_step']['ms_per_step'],4), round(l['roofline']['avg_launch_ms']*1e3,2), l['parity']['clips_with_identical_text'], l['parity']['words_identical_and_within_20ms'])"
done; done
