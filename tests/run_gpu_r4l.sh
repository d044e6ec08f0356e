#!/bin/bash
export PYTHONUNBUFFERED=1
for d in 0 1; do
CW_MEL_DBG=$d python - <<PY
import sys, time
sys.path.insert(0, '.')
import numpy as np
from crisperwhisper_amd import synthetic as syn
from crisperwhisper_amd.engine import Engine
from tests import helpers as Hh
g, v, W, spec = Hh.tiny_setup()
eng = Engine(spec, dtype="bf16", max_batch=8)
clips = [syn.synth_audio(i, 480000, "noise") for i in range(8)]
eng.upload_pcm(clips)
for _ in range(3): eng.mel_resident(8)
eng.sync(); eng.stage_times(reset=True)
for _ in range(20): eng.mel_resident(8)
eng.sync()
st = eng.stage_times()
print("dbg=$d mel stage ms/call", st["mel"][0] / st["mel"][1])
eng.close()
PY
done
