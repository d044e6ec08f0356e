"""Host half of beam search: the native bookkeeping (csrc/beamhost.cpp, one C call per decoder step) against the numpy
statement of GenerationMixin._beam_search in generation.beam_search(native_host=False) -- itself pinned to transformers by the
beam goldens (tests/test_pipeline_cpu.py, tests/test_gpu_e2e.py).  Both are driven by the SAME pre-drawn candidate streams
(what cw_beam_step would hand over: per row the best 2K processed log-probabilities, best first, -inf / -1 padded); every
(parent, token) pair handed to cw_beam_advance, the step count and the returned sequences, beam indices, scores and alignment
row map must be identical, bit for bit.  Values are drawn on a coarse grid so that accumulated scores tie across beams and
the (value desc, flattened index asc) order of torch.topk is exercised.  Host-only: no GPU."""
import types

import numpy as np
import pytest

from crisperwhisper_amd import generation


class CandidateStream:
    """Stands in for the device half: replays pre-drawn candidates, records what the host half asks for."""

    def __init__(self, seed, B, K, vocab, eos, pad, p_eos, grid, n_steps):
        self.spec = types.SimpleNamespace(vocab_size=vocab, eos_token_id=eos, pad_token_id=pad)
        rng = np.random.default_rng(seed)
        self.steps = []
        rows, keep = B * K, 2 * K
        for _ in range(n_steps):
            v = -np.sort(rng.integers(0, 40, (rows, keep)) * grid + rng.integers(0, 2, (rows, 1)) * grid)[:, ::-1]
            v = np.sort(v.astype(np.float32), axis=1)[:, ::-1].copy()            # best first
            t = np.stack([rng.choice(vocab, keep, replace=False) for _ in range(rows)]).astype(np.int32)
            force = rng.random((rows, keep)) < p_eos                             # sprinkle eos candidates
            for r in range(rows):
                if force[r].any():
                    j = int(np.argmax(force[r]))
                    if eos not in t[r]:
                        t[r, j] = eos
            npad = rng.integers(0, keep, rows) * (rng.random(rows) < 0.15)       # some rows offer fewer allowed tokens
            for r in range(rows):
                if npad[r]:
                    v[r, keep - npad[r]:] = -np.inf
                    t[r, keep - npad[r]:] = -1
            self.steps.append((v, t))
        self.reset()

    def reset(self):
        self.t = 0
        self.advances = []
        self.finish = None

    def beam_begin(self, prompt, K, max_length, min_new_tokens):
        self.reset()

    def beam_step(self, keep):
        v, t = self.steps[self.t]
        assert v.shape[1] == keep
        self.t += 1
        return v.copy(), t.copy()

    def beam_advance(self, parent, token):
        self.advances.append((np.array(parent, np.int64).copy(), np.array(token, np.int64).copy()))

    def beam_finish(self, unrolled):
        self.finish = np.array(unrolled).copy()


CASES = [  # B, K, n_prompt, max_length, vocab, eos, pad, p_eos, length_penalty, early_stopping
    (1, 1, 3, 12, 50, 7, None, 0.10, 1.0, False),
    (2, 2, 3, 20, 64, 5, 5, 0.05, 1.0, False),
    (3, 5, 4, 40, 300, 11, None, 0.03, 1.0, False),
    (8, 5, 4, 60, 51866, 50257, 50257, 0.02, 1.0, False),
    (3, 5, 4, 40, 300, 11, 0, 0.03, 0.6, False),
    (2, 3, 2, 30, 100, 9, 3, 0.08, 2.0, True),
    (4, 4, 3, 25, 80, 2, None, 0.30, 1.0, True),
    (2, 5, 3, 9, 90, 4, None, 0.00, 1.0, False),      # nobody offers eos: the search ends at max_length
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_native_host_bookkeeping_is_bit_equal_to_the_numpy_statement(case, seed):
    B, K, n_prompt, max_length, vocab, eos, pad, p_eos, lp, es = case
    eng = CandidateStream(100 * seed + B + K, B, K, vocab, eos, pad, p_eos, np.float32(0.25), max_length)
    prompt = np.random.default_rng(seed).integers(0, vocab, (B, n_prompt)).astype(np.int64)
    out = {}
    for native in (False, True):
        res = generation.beam_search(eng, prompt, max_length, 0, K, length_penalty=lp, early_stopping=es, native_host=native)
        out[native] = (res, list(eng.advances), eng.finish.copy(), eng.t)
    (r0, a0, f0, t0), (r1, a1, f1, t1) = out[False], out[True]
    assert t0 == t1 and len(a0) == len(a1)
    for (p0, k0), (p1, k1) in zip(a0, a1):
        assert np.array_equal(p0, p1) and np.array_equal(k0, k1)
    assert np.array_equal(r0[0], r1[0]) and r0[0].dtype == r1[0].dtype          # sequences
    assert np.array_equal(r0[1], r1[1])                                           # beam indices
    assert r0[2] == r1[2]                                                         # alignment rows gathered
    assert np.array_equal(r0[3].view(np.uint32), r1[3].view(np.uint32))           # scores, bit for bit
    assert np.array_equal(f0, f1)


def test_native_host_rejects_bad_geometry():
    eng = CandidateStream(0, 1, 2, 20, 3, None, 0.1, np.float32(0.5), 4)
    with pytest.raises(ValueError):
        generation.beam_search(eng, np.zeros((1, 5), np.int64), 5, 0, 2)          # max_length == prompt length
