"""Output writers for the pipeline result (consumer side of REF/app.py:74-82)."""
from __future__ import annotations

import json
from typing import Dict, List


def _vtt_time(t: float) -> str:
    """Same formatting as REF/app.py:79-80: H:MM:SS.mmm, seconds rounded to milliseconds by the format."""
    return f"{int(t // 3600)}:{int(t // 60 % 60):02d}:{t % 60:06.3f}"


def timestamps_to_vtt(chunks: List[Dict]) -> str:
    """WEBVTT cue per word, like ``timestamps_to_vtt`` in REF/app.py:74-82."""
    out = "WEBVTT\n\n"
    for w in chunks:
        start, end = w["timestamp"]
        out += f"{_vtt_time(start)} --> {_vtt_time(end)}\n{w['text']}\n\n"
    return out


def timestamps_to_srt(chunks: List[Dict]) -> str:
    out = []
    for i, w in enumerate(chunks, 1):
        s, e = (_vtt_time(t).replace(".", ",") for t in w["timestamp"])
        out.append(f"{i}\n{s} --> {e}\n{w['text'].strip()}\n")
    return "\n".join(out)


def to_json(result: Dict) -> str:
    return json.dumps({"text": result["text"], "chunks": [{"text": c["text"], "timestamp": list(c["timestamp"])}
                                                         for c in result["chunks"]]}, ensure_ascii=False)
