"""Where the 16-bit engines part from the reference on the 600 s recording (BASELINE configs[2], bench geometry), and by how much.

The f32 engine reproduces the transformers output of this recording word for word (tests/test_gpu_e2e.py::
test_longform_600s_at_bench_geometry_vs_transformers[f32]), so its token sequences ARE the reference's.  The recording's 30 windows
are decoded in batches of 4 with the host seek loop (generation.generate(native=False)) on the f32 engine and on a 16-bit engine;
every decoder call is recorded.  For the first position at which a row of a call differs, both engines are run teacher-forced on the
reference tokens of that call and the logits of that position are captured: the report lists the reference token, the token the
16-bit engine chose, the f32 engine's margin between them and the 16-bit engine's deviation at those two logits.  The choice is
made on PROCESSED scores (suppress lists, timestamp grammar, "timestamp mass beats the best text token" rule: oracle/logits.py,
the checker's restatement of TF logits_process.py:1851-2047), so the margin of the rule that decided is reported as well:
`timestamp_rule_margin` = logsumexp(log p[timestamps]) - max(log p[text]).  A divergence whose margin is below the 16-bit
engine's own logit noise is a tie broken by rounding, not an error in the path.  (Checker-side script: lives under tests/.)

    python tests/longform_divergence.py [bf16|f16 ...]  ->  gpurun_out/longform_divergence.json (+ table on stdout)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crisperwhisper_amd import audio as cw_audio, generation, synthetic as syn
from crisperwhisper_amd.engine import Engine
from oracle import logits as OL


class Recorder:
    """Wraps an Engine: records every encode (items, seek) and decode (prompt, sequences, lengths) of the host seek loop."""

    def __init__(self, eng):
        self.eng, self.calls, self._enc = eng, [], None

    def __getattr__(self, name):
        return getattr(self.eng, name)

    def encode(self, items, seek, seek_num):
        self._enc = (list(items), np.asarray(seek).copy(), np.asarray(seek_num).copy())
        return self.eng.encode(items, seek, seek_num)

    def decode(self, prompt, max_length, min_new_tokens=0, **kw):
        out = self.eng.decode(prompt, max_length, min_new_tokens, **kw)
        self.calls.append({"enc": self._enc, "prompt": np.asarray(prompt).copy(), "max_length": int(max_length),
                           "seqs": np.asarray(out[0]).copy(), "lens": np.asarray(out[1]).copy()})
        return out


def run(eng, windows, pcm, gk, batch):
    rec = Recorder(eng)
    per_batch = []
    for b0 in range(0, len(windows), batch):
        idxs = list(range(b0, min(b0 + batch, len(windows))))
        clips = [pcm[windows[i][0]: windows[i][0] + windows[i][1]] for i in idxs]
        _, nf = eng.mel(clips)
        n0 = len(rec.calls)
        generation.generate(rec, len(idxs), nf, language=gk["language"], task=gk["task"], max_new_tokens=gk["max_new_tokens"],
                            min_new_tokens=gk["min_new_tokens"], num_beams=1, native=False)
        per_batch.append((idxs, clips, rec.calls[n0:]))
    return per_batch


def main():
    dtypes = sys.argv[1:] or ["bf16", "f16"]
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "e2e_bench_longform_golden.json")))
    g, v = syn.large_v3_geometry()
    spec = syn.model_spec(g, v, n_align=15)
    a = gold["audio"]
    pcm = syn.synth_audio(a["seed"], a["secs"] * 16000, a["kind"])
    gk = gold["generate_kwargs"]
    batch = gold["pipeline"]["batch_size"]
    windows = cw_audio.chunk_windows(len(pcm), 480000, 80000, 80000)
    W = {n: syn.weight_tensor(g, n, s, gold["weight_seed"], "aligned") for n, s in syn.weight_shapes(g).items()}
    e32 = Engine(spec, dtype="f32", max_batch=batch)
    e32.load_state_dict(W)
    ref = run(e32, windows, pcm, gk, batch)
    report = {"recording": a, "windows": len(windows), "batch": batch, "engines": {}}
    for dt in dtypes:
        e16 = Engine(spec, dtype=dt, max_batch=batch)
        e16.load_state_dict(W)
        got = run(e16, windows, pcm, gk, batch)
        rows_total = rows_diff = 0
        events = []
        for (idxs, clips, rc), (_, _, gc) in zip(ref, got):
            for ci, (cr, cg) in enumerate(zip(rc, gc)):
                if cr["enc"][0] != cg["enc"][0] or not np.array_equal(cr["enc"][1], cg["enc"][1]):
                    break                                     # an earlier divergence moved the seek positions: later calls are not comparable
                n_prompt = cr["prompt"].shape[1]
                for row in range(cr["seqs"].shape[0]):
                    rows_total += 1
                    L = int(min(cr["lens"][row], cg["lens"][row]))
                    d = np.nonzero(cr["seqs"][row, :L] != cg["seqs"][row, :L])[0]
                    if len(d) == 0 and cr["lens"][row] == cg["lens"][row]:
                        continue
                    rows_diff += 1
                    pos = int(d[0]) if len(d) else L
                    # teacher-forced logits of position `pos` on both engines (reference tokens as the forced continuation)
                    T = pos + 1
                    forced = np.full(cr["seqs"].shape, -1, np.int32)[:, :max(T, n_prompt + 1)]
                    forced[:, n_prompt:T] = cr["seqs"][:, n_prompt:T]
                    logits = {}
                    for name, eng in (("f32", e32), (dt, e16)):
                        eng.mel(clips)
                        items, seek, seek_num = cr["enc"]
                        eng.encode(items, seek, seek_num)
                        cap = eng.capture_logits(len(items), T)
                        eng.decode(cr["prompt"], max_length=T, forced=forced[:, :T])
                        logits[name] = cap[T - 1 - n_prompt, row].copy() if T > n_prompt else None   # logits that choose token `pos`
                        eng.stop_capture()
                    t_ref, t_got = int(cr["seqs"][row, pos]), int(cg["seqs"][row, pos])
                    l32, l16 = logits["f32"], logits[dt]
                    ps = OL.ProcessorSpec(eos=spec.eos_token_id, no_timestamps=spec.no_timestamps_token_id, suppress=spec.suppress_tokens,
                                          begin_suppress=spec.begin_suppress_tokens, max_initial_timestamp_index=spec.max_initial_timestamp_index,
                                          min_new_tokens=gk["min_new_tokens"])
                    prefix = cr["seqs"][row: row + 1, :pos].astype(np.int64)
                    proc = {name: OL.process(ps, prefix, lg[None, :], n_prompt, n_prompt)[0] for name, lg in (("f32", l32), (dt, l16))}
                    mass = {name: float(OL.timestamp_mass_margin(ps, prefix, lg[None, :], n_prompt, n_prompt)[0]) for name, lg in (("f32", l32), (dt, l16))}
                    assert int(np.argmax(proc["f32"])) == t_ref, "the f32 engine's processed argmax is the reference token"
                    both = (mass["f32"] > 0) == (mass[dt] > 0)           # the mass rule fires on both engines or on neither
                    ev = {"window": idxs[cr["enc"][0][row]] if row < len(cr["enc"][0]) else None, "call": ci, "position": pos, "reference_token": t_ref,
                          "engine_token": t_got, "f32_margin_ref_minus_engine_token": float(l32[t_ref] - l32[t_got]),
                          "decided_by": "argmax among tokens both allowed" if both else "timestamp-mass rule (logsumexp of timestamps vs best text token)",
                          "timestamp_rule_margin_f32": mass["f32"], "timestamp_rule_margin_engine": mass[dt],
                          "f32_top1_minus_top2": float(np.sort(l32)[-1] - np.sort(l32)[-2]),
                          "engine_logit_error_at_the_two_tokens": [float(l16[t_ref] - l32[t_ref]), float(l16[t_got] - l32[t_got])],
                          "engine_logit_rms_error_over_vocabulary": float(np.sqrt(((l16 - l32) ** 2).mean())),
                          "f32_logit_range": float(l32.max() - l32.min())}
                    events.append(ev)
        report["engines"][dt] = {"decoder_rows_compared": rows_total, "rows_that_differ_first_divergence_only": rows_diff, "divergences": events}
        e16.close()
        print(f"== {dt}: {rows_diff} of {rows_total} comparable decoder rows diverge")
        for ev in events:
            print("  window %2s call %d pos %3d  ref %5d engine %5d  f32 margin %+.4f  (top1-top2 %.4f)  engine error at the two tokens %+.4f %+.4f  rms %.4f  range %.1f  | %s: mass margin f32 %+.4f engine %+.4f"
                  % (ev["window"], ev["call"], ev["position"], ev["reference_token"], ev["engine_token"], ev["f32_margin_ref_minus_engine_token"],
                     ev["f32_top1_minus_top2"], *ev["engine_logit_error_at_the_two_tokens"], ev["engine_logit_rms_error_over_vocabulary"], ev["f32_logit_range"],
                     "argmax" if ev["decided_by"].startswith("argmax") else "MASS RULE", ev["timestamp_rule_margin_f32"], ev["timestamp_rule_margin_engine"]))
    e32.close()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(report, open("gpurun_out/longform_divergence.json", "w"), indent=1)


if __name__ == "__main__":
    main()
