"""Pause splitting, restatement of REF/utils.py:1-29 (test oracle)."""


def adjust_pauses_for_hf_pipeline_output(pipeline_output, split_threshold=0.12):
    chunks = pipeline_output["chunks"].copy()          # shallow copy: dicts are mutated (REF/utils.py:6)
    for i in range(len(chunks) - 1):
        cs, ce = chunks[i]["timestamp"]
        ns, ne = chunks[i + 1]["timestamp"]
        pause = ns - ce
        if pause > 0:
            d = split_threshold / 2 if pause > split_threshold else pause / 2
            chunks[i]["timestamp"] = (cs, ce + d)
            chunks[i + 1]["timestamp"] = (ns - d, ne)
    pipeline_output["chunks"] = chunks
    return pipeline_output
