"""Which e4m3 pieces of the opt-in fp8 mode (BASELINE configs[3]) move the transcript: batch 64 x 30 s at the bench geometry,
every clip compared with its transformers reference (tests/golden/e2e_bench_golden.json + e2e_bench_b64_golden.json), each
e4m3 GEMM of the encoder alone, cumulatively, and the e4m3 cross-attention cache alone / on top.
Mask bits of `encoder_gemm_fp8`: 1 q/k/v, 2 fc1, 4 fc2, 8 cross-K/V projection.

    python tools/fp8_sweep.py  ->  table on stdout + gpurun_out/fp8_sweep.json (copied to profiles/r04_fp8_sweep.*)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crisperwhisper_amd import collate, generation, synthetic as syn
from crisperwhisper_amd.engine import Engine

B = 64


def goldens():
    out = {}
    for name in ("e2e_bench_golden.json", "e2e_bench_b64_golden.json"):
        gj = json.load(open(os.path.join(ROOT, "tests", "golden", name)))
        for c in gj["clips"]:
            out.setdefault(int(c["seed"]), c)
    return out, gj["generate_kwargs"]


def main():
    gold, gk = goldens()
    g, v = syn.large_v3_geometry()
    spec = syn.model_spec(g, v, n_align=15)
    vocab = collate.Vocabulary.from_synthetic(v)
    clips = [syn.synth_audio(i, 480000, "noise") for i in range(B)]
    W = {n: syn.weight_tensor(g, n, s, 0, "aligned") for n, s in syn.weight_shapes(g).items()}
    names = {1: "q/k/v", 2: "fc1", 4: "fc2", 8: "cross-K/V proj"}
    configs = [(0, None)] + [(m, None) for m in (1, 2, 4, 8, 6, 3, 5, 7, 14, 15)] + [(0, "fp8"), (2, "fp8"), (6, "fp8"), (14, "fp8"), (15, "fp8")]
    rows = []
    for mask, kv in configs:
        eng = Engine(spec, dtype="bf16", max_batch=B, cross_kv_dtype=kv)
        try:
            eng.load_state_dict(W)
            if mask:
                eng.check_weights()
                eng.set_encoder_gemm_fp8(True, mask=int(mask))
            nf = eng.upload_pcm(clips)

            def one():
                eng.mel_resident(B)
                return generation.generate(eng, B, nf, language=gk["language"], task=gk["task"], max_new_tokens=gk["max_new_tokens"],
                                           min_new_tokens=gk["min_new_tokens"])
            one()
            eng.stage_times(reset=True)
            eng.sync()
            t0 = time.perf_counter()
            res = one()
            eng.sync()
            dt = time.perf_counter() - t0
            st = eng.stage_times()
            same = ok = tot = 0
            differing = []
            for k in range(B):
                n = len(res["token_timestamps"][k])
                text, ws = collate.decode_asr(vocab, [{"tokens": res["sequences"][k][:n], "token_timestamps": res["token_timestamps"][k], "stride": (30.0, 0.0, 0.0)}])
                ref = gold[k]
                if text == ref["text"] and len(ws) == len(ref["chunks"]):
                    same += 1
                    for wa, wb in zip(ws, ref["chunks"]):
                        tot += 1
                        ok += int(wa["text"] == wb["text"] and all(abs(x - y) <= 0.02 + 1e-9 for x, y in zip(wa["timestamp"], wb["timestamp"])))
                else:
                    differing.append(k)
            label = " + ".join(names[b] for b in (1, 2, 4, 8) if mask & b) or "none"
            row = {"encoder_gemm_fp8_mask": mask, "e4m3_gemms": label, "cross_kv_cache": kv or "bf16", "clips_identical_text": same, "of": B,
                   "words_within_20ms": [ok, tot], "differing": differing, "ms_per_step": dt * 1e3,
                   "encoder_ms": st["encoder"][0] / max(st["encoder"][1], 1) * (st["encoder"][1]), "cross_kv_ms": st["cross_kv"][0], "decode_ms": st["decode"][0]}
            rows.append(row)
            print(f"mask {mask:2d} ({label:38s}) cache {row['cross_kv_cache']:4s}: {same:2d}/{B} clips identical, words {ok}/{tot}, step {dt * 1e3:7.1f} ms "
                  f"(encoder {row['encoder_ms']:6.1f}, cross-K/V {row['cross_kv_ms']:5.1f}, decode {row['decode_ms']:6.1f})  differing {differing}", flush=True)
        finally:
            eng.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "fp8_sweep.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
