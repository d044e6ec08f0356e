"""crisperwhisper_amd -- MI355X-native CrisperWhisper inference-and-alignment path.

Public surface (mirrors what REF/transcribe.py, REF/app.py and REF/utils.py use):
    pipeline(...)                                  -> CrisperWhisperPipeline (HF ASR pipeline protocol)
    adjust_pauses_for_hf_pipeline_output(out, thr) -> same dict, pauses redistributed (on device)
"""
from .collate import Vocabulary
from .engine import Engine, EngineError, ModelSpec
from .pipeline import CrisperWhisperPipeline, ModelBundle, pipeline
from .utils import adjust_pauses_for_hf_pipeline_output

__all__ = ["pipeline", "CrisperWhisperPipeline", "ModelBundle", "ModelSpec", "Engine", "EngineError", "Vocabulary",
           "adjust_pauses_for_hf_pipeline_output"]
