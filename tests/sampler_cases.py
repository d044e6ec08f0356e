"""Crafted + random rows for the differential tests of the logits-processor stage.

Every case is (ids [t] = prompt + generated so far, logits [V], min_new_tokens); the expected token is whatever
HF's processor list + argmax yields (tests/test_oracle_vs_golden.py pins oracle/logits.py to it, tests/test_gpu_kernels.py
pins the HIP sample_kernel to the oracle).  Vocabulary layout: crisperwhisper_amd.synthetic.SynthVocab."""
from __future__ import annotations

import numpy as np

from crisperwhisper_amd import synthetic as syn

N_PROMPT = 3


def cases(v: syn.SynthVocab, V: int, seed: int = 0):
    rng = np.random.default_rng(seed)
    tb, eos = v.timestamp_begin, v.eos
    prompt = [v.sot, v.lang_id("en"), v.transcribe]
    A, Bt = ord("a"), ord("b")                     # unsuppressed text bytes
    out = []

    def base(scale=1.0):
        return (rng.standard_normal(V) * scale).astype(np.float32)

    def add(name, gen, lg, min_new=0):
        out.append((name, np.asarray(prompt + list(gen), np.int64), lg.astype(np.float32), int(min_new)))

    # --- first generated token: must be a timestamp <= tb + max_initial (50), begin-suppress list active
    lg = base(); lg[A] = 50.0
    add("begin_text_best_is_masked", [], lg)
    lg = base(); lg[tb + 51] = 60.0; lg[tb + 50] = 10.0; lg[tb + 7] = 9.0
    add("begin_beyond_max_initial_is_masked", [], lg)
    lg = base(); lg[tb + 3] = lg[tb + 9] = 20.0
    add("begin_tie_between_timestamps_lowest_index", [], lg)
    # --- after (text, ts): only timestamps / >= eos
    lg = base(); lg[A] = 30.0; lg[tb + 12] = 5.0
    add("after_text_ts_text_is_masked", [tb, A, tb + 10], lg)
    lg = base(); lg[A] = 30.0; lg[eos] = 25.0; lg[tb:] = -20.0
    add("after_text_ts_eos_wins", [tb, A, tb + 10], lg)
    lg = base(); lg[eos] = 25.0; lg[tb:] = -20.0
    add("after_text_ts_eos_masked_by_min_new_tokens", [tb, A, tb + 10], lg, min_new=10)
    lg = base(); lg[tb + 10] = 9.0; lg[tb + 9] = 30.0
    add("closing_timestamp_may_repeat_but_not_decrease", [tb, A, tb + 10], lg)
    # --- after (ts, ts): text only
    lg = base(); lg[tb + 40] = 30.0; lg[A] = 1.0
    add("after_ts_ts_timestamps_masked", [tb, A, tb + 10, tb + 10], lg)
    # --- monotonicity after a closed pair: timestamps <= last are masked (strictly increasing)
    lg = base(); lg[tb + 10] = 40.0; lg[tb + 11] = 39.0; lg[:tb] = -30.0
    add("after_pair_timestamp_must_increase", [tb, A, tb + 10, tb + 10, Bt], lg)
    # --- logsumexp rule
    lg = np.full(V, -30.0, np.float32); lg[A] = 5.0; lg[tb + 20: tb + 30] = 3.0      # 10 * e^3 > e^5
    add("logsumexp_forces_timestamp", [tb, A], lg)
    lg = np.full(V, -30.0, np.float32); lg[A] = 5.0; lg[tb + 20: tb + 22] = 3.0      # 2 * e^3 < e^5
    add("logsumexp_keeps_text", [tb, A], lg)
    lg = np.full(V, -np.inf, np.float32); lg[A] = 4.0; lg[tb + 20] = 4.0             # logsumexp == max text: not '>'
    add("logsumexp_equal_is_not_greater_tie_to_text", [tb, A], lg)
    lg = np.full(V, -np.inf, np.float32); lg[Bt] = 4.0; lg[A] = 4.0
    add("text_tie_lowest_index", [tb, A], lg)
    # --- degenerate rows
    lg = np.full(V, -np.inf, np.float32)
    add("all_minus_inf", [tb, A], lg)
    lg = np.full(V, -np.inf, np.float32); lg[tb + 5] = 1.0
    add("only_a_masked_timestamp_is_finite", [tb, A, tb + 10, tb + 10, Bt], lg)         # tb+5 < last+1 -> masked too
    lg = base(); lg[v.notimestamps] = 99.0; lg[v.sot] = 98.0; lg[ord("(")] = 97.0
    add("suppressed_tokens", [tb, A], lg)
    lg = base(); lg[ord(" ")] = 99.0
    add("begin_suppress_only_at_begin", [tb], lg)
    # --- random rows over random grammar states
    for i in range(40):
        n = int(rng.integers(0, 12))
        gen, last_ts = [], tb
        for k in range(n):
            if k == 0:
                tok = tb + int(rng.integers(0, 5))
            elif gen[-1] >= tb and (len(gen) < 2 or gen[-2] >= tb):
                tok = int(rng.integers(97, 123))
            elif gen[-1] >= tb:
                tok = gen[-1] if rng.random() < 0.5 else gen[-1] + int(rng.integers(1, 20))
            else:
                tok = int(rng.integers(97, 123)) if rng.random() < 0.6 else last_ts + int(rng.integers(1, 30))
            if tok >= tb:
                last_ts = tok
            gen.append(tok)
        lg = base(3.0)
        if i % 3 == 0:
            lg[tb:] += 2.0
        if i % 4 == 0:
            lg[rng.integers(0, V, 50)] = -np.inf
        add(f"random{i}", gen, lg, min_new=int(rng.integers(0, 3)) * 4)
    return out
