"""Controlled timing of the beam-search decoder step: ONE generate pass (8 items x 5 hypotheses = 40 decoder rows, 128 forced-length
tokens) over freshly encoded bench clips, repeated -- every kernel variant sees the same rows and the same number of steps, unlike
`bench.py --num-beams 5`, whose seek loop runs a data-dependent number of passes (a variant that moves one near-tie can add a third,
cheaper pass and bias the per-step average).  Prints ms per beam step (host wall incl. the host half of the search) and a digest of
the sequences.   usage: python tools/beam_step_bench.py [--items 8] [--beams 5] [--tokens 128] [--reps 3] [--cross-kv fp8]"""
import argparse, hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crisperwhisper_amd import generation, synthetic as syn
from crisperwhisper_amd.engine import Engine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=8); ap.add_argument("--beams", type=int, default=5)
    ap.add_argument("--tokens", type=int, default=128); ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--dtype", default="bf16"); ap.add_argument("--cross-kv", default=None)
    a = ap.parse_args()
    g, v = syn.large_v3_geometry()
    spec = syn.model_spec(g, v, n_align=15)
    eng = Engine(spec, dtype=a.dtype, max_batch=a.items * a.beams, cross_kv_dtype=a.cross_kv)
    for name, shape in syn.weight_shapes(g).items():
        eng.load_tensor(name, syn.weight_tensor(g, name, shape, 0, "aligned"))
    clips = [syn.synth_audio(i, 480000, "noise") for i in range(a.items)]
    eng.mel(clips)
    eng.encode(list(range(a.items)), [0] * a.items, [3000] * a.items)
    prompt = np.tile(np.array([[v.sot, v.lang_id("en"), v.transcribe]], np.int32), (a.items, 1))
    T = 3 + a.tokens
    out = None
    times = []
    for r in range(a.reps + 1):
        eng.sync(); t0 = time.perf_counter()
        out = generation.beam_search(eng, prompt, T, a.tokens, a.beams)
        eng.sync(); times.append(time.perf_counter() - t0)
    seqs = np.asarray(out[0])
    print(f"beam_step_bench items={a.items} beams={a.beams} tokens={a.tokens} cross_kv={a.cross_kv}: "
          f"ms per beam step {min(times[1:]) / a.tokens * 1e3:.4f} (best of {a.reps}; all {[round(t / a.tokens * 1e3, 4) for t in times[1:]]}), "
          f"sequences sha1 {hashlib.sha1(seqs.tobytes()).hexdigest()[:12]}")
    eng.close()


if __name__ == "__main__":
    main()
