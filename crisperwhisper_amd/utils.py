"""``adjust_pauses_for_hf_pipeline_output`` -- same name, arguments and in-place behaviour as
REF/utils.py:1-29, with the per-boundary arithmetic executed on device (``cw_adjust_pauses``)."""
from __future__ import annotations

import numpy as np

_engine = None


def bind_engine(engine):
    """Called by the pipeline: pause splitting runs on the same context/GPU."""
    global _engine
    _engine = engine


def adjust_pauses_for_hf_pipeline_output(pipeline_output, split_threshold=0.12, engine=None):
    """Adjust pause timings by distributing pauses up to the threshold evenly between adjacent words.
    Mutates and returns ``pipeline_output`` exactly like the reference (the chunk dicts are updated in place)."""
    eng = engine or _engine
    if eng is None:
        raise RuntimeError("no device engine bound: build a crisperwhisper_amd.pipeline(...) first or pass engine=; "
                           "there is no CPU fallback")
    chunks = pipeline_output["chunks"].copy()
    if chunks:
        start = np.array([c["timestamp"][0] for c in chunks], dtype=np.float64)
        end = np.array([c["timestamp"][1] for c in chunks], dtype=np.float64)
        s2, e2 = eng.adjust_pauses(start, end, float(split_threshold))
        for c, a, b, a0, b0 in zip(chunks, s2, e2, start, end):
            if a != a0 or b != b0:
                c["timestamp"] = (float(a), float(b))
    pipeline_output["chunks"] = chunks
    return pipeline_output
