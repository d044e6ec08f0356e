"""Shared test scaffolding (test infrastructure: may import oracle/)."""
from __future__ import annotations

import json
import os

import numpy as np

from crisperwhisper_amd import collate, synthetic as syn
from oracle import collate as OC
from oracle import generate as OG
from oracle import timestamps as OT
from oracle.model import WhisperOracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GEN_KW = {"num_beams": 1, "language": "<|en|>", "task": "transcribe"}


def gold_npz(name):
    return np.load(os.path.join(GOLD, name))


def gold_json(name):
    return json.load(open(os.path.join(GOLD, name)))


def tiny_setup(seed=0, n_align=3):
    g, v = syn.tiny_geometry()
    W = syn.random_weights(g, seed=seed)
    spec = syn.model_spec(g, v, n_align)
    return g, v, W, spec


def oracle_spec(g, v, spec):
    return OG.GenSpec(eos=v.eos, pad=v.eos, sot=v.sot, no_timestamps=v.notimestamps, lang_to_id=spec.lang_to_id,
                      task_to_id=spec.task_to_id, alignment_heads=[list(h) for h in spec.alignment_heads],
                      suppress=v.suppress_tokens(), begin_suppress=v.begin_suppress_tokens(),
                      max_initial_timestamp_index=50, max_length=g.max_target_positions,
                      median_filter_width=g.median_filter_width)


def oracle_vocab(v):
    return OC.ByteVocab(v.token_bytes(), v.special_names(), v.eos, v.timestamp_begin, v.startofprev, v.sot)


def words_equal(got, want, tol=0.0):
    """word-for-word text equality, |dt| <= tol."""
    if len(got) != len(want):
        return False, f"{len(got)} words vs {len(want)}"
    for i, (a, b) in enumerate(zip(got, want)):
        if a["text"] != b["text"]:
            return False, f"word {i}: {a['text']!r} vs {b['text']!r}"
        for x, y in zip(a["timestamp"], b["timestamp"]):
            if abs(x - y) > tol + 1e-9:
                return False, f"word {i} ({a['text']!r}): {a['timestamp']} vs {b['timestamp']}"
    return True, ""


class OracleBackedEngine:
    """Stands in for crisperwhisper_amd.engine.Engine in CPU tests of the *host* control flow
    (generation.generate / pipeline): same method contract, arithmetic done by the oracle."""

    def __init__(self, g, v, W, spec):
        from oracle import mel as OM
        self.OM = OM
        self.spec = spec
        self.g = g
        self.model = WhisperOracle(W, g)
        self.ospec = oracle_spec(g, v, spec)
        self.max_batch = 64

    def mel(self, clips, return_features=False):
        pcm = np.zeros((len(clips), self.OM.N_SAMPLES), np.float32)
        nf = np.zeros(len(clips), np.int32)
        for i, c in enumerate(clips):
            pcm[i], nv = self.OM.pad_or_trim(c)
            nf[i] = self.OM.attention_mask_frames(nv)
        self.feats = self.OM.log_mel(pcm, self.g.n_mels)
        return (self.feats if return_features else None), nf

    def encode(self, item, seek, n_frames):
        seg = np.zeros((len(item), self.g.n_mels, 3000), np.float32)
        for i, (it, s, n) in enumerate(zip(item, seek, n_frames)):
            seg[i, :, :n] = self.feats[it, :, s:s + n]
        self.enc = self.model.encode(seg)

    def decode(self, prompt, max_length, min_new_tokens=0, forced=None, want_argmax=False):
        prompt = np.asarray(prompt, dtype=np.int64)
        self._last_prompt = prompt
        n_prompt = prompt.shape[1]
        seqs, weights = OG.greedy(self.model, self.ospec, self.enc, prompt, begin_index=n_prompt,
                                  max_new_tokens=max_length - n_prompt, min_new_tokens=min_new_tokens)
        self.weights = weights
        self._last_logits = getattr(self.model, "_dbg_last_logits", None)
        out = np.full((seqs.shape[0], self.spec.max_target_positions), self.spec.pad_token_id, np.int32)
        out[:, :seqs.shape[1]] = seqs
        lens = np.full(seqs.shape[0], seqs.shape[1], np.int32)
        return out, lens, None

    def last_logits(self, nb):
        """logits after feeding the prompt (used by language detection: prompt = <|startoftranscript|>)."""
        cache = self.model.new_cache(self.enc)
        logits, _ = self.model.decode(self._last_prompt, cache)
        return logits[:nb]

    def token_timestamps(self, nb, L, n_prompt, num_frames):
        assert self.weights.shape[2] == L
        return OT.extract_token_timestamps(self.weights, num_frames, n_prompt, self.g.median_filter_width)

    # ---- beam search: same contract as Engine.beam_begin / beam_step / beam_advance / beam_finish
    def beam_begin(self, prompt, num_beams, max_length, min_new_tokens=0):
        from oracle import logits as OL
        prompt = np.asarray(prompt, dtype=np.int64)
        self._bk = int(num_beams)
        self._b_ids = np.repeat(prompt, self._bk, axis=0)
        self._b_nprompt = prompt.shape[1]
        self._b_cache = self.model.new_cache(np.repeat(self.enc[:prompt.shape[0]], self._bk, axis=0))
        self._b_pending = self._b_ids.copy()
        self._b_weights = []
        self._b_spec = OL.ProcessorSpec(eos=self.ospec.eos, no_timestamps=self.ospec.no_timestamps, suppress=self.ospec.suppress,
                                        begin_suppress=self.ospec.begin_suppress,
                                        max_initial_timestamp_index=self.ospec.max_initial_timestamp_index,
                                        min_new_tokens=min_new_tokens)

    def beam_step(self, n_cand):
        from oracle import logits as OL
        logits, cross = self.model.decode(self._b_pending, self._b_cache, want_heads=self.ospec.alignment_heads)
        self._b_weights.append(cross)
        with np.errstate(invalid="ignore"):
            lp = OL.process(self._b_spec, self._b_ids, OL.log_softmax(logits), self._b_nprompt, self._b_nprompt)
        order = np.lexsort((np.broadcast_to(np.arange(lp.shape[1]), lp.shape), -lp), axis=-1)[:, :n_cand]
        vals = np.take_along_axis(lp, order, axis=1).astype(np.float32)
        toks = np.where(np.isfinite(vals), order, -1).astype(np.int32)
        return vals, toks

    def beam_advance(self, parent, token):
        parent = np.asarray(parent, dtype=np.int64)
        for key in ("self_k", "self_v"):
            self._b_cache[key] = [a[parent] for a in self._b_cache[key]]
        self._b_ids = np.concatenate([self._b_ids[parent], np.asarray(token, dtype=np.int64)[:, None]], axis=1)
        self._b_pending = self._b_ids[:, -1:]

    def beam_finish(self, row_of_pos):
        w = np.concatenate(self._b_weights, axis=2)                   # [rows, Ha, positions, S]
        r = np.asarray(row_of_pos)
        out = np.stack([np.stack([w[r[i, p], :, p, :] for p in range(r.shape[1])], axis=1) for i in range(r.shape[0])])
        self.weights = out.astype(np.float32)

    def adjust_pauses(self, start, end, thr):
        raise AssertionError("pause splitting must run on the device in product code")


def has_experiments() -> bool:
    """The library was built with `make EXTRA=-DCW_EXPERIMENTS`: the measured-and-rejected kernel variants ("A/B
    switches") and the differential tests that hold them correct are available."""
    from crisperwhisper_amd import _native
    return bool(_native.load().cw_has_experiments())


def e4m3_round(x):
    """round-to-nearest-even onto the OCP e4m3 grid (|x| <= 448 saturating), numpy float64 in / out"""
    x = np.asarray(x, np.float64)
    a = np.abs(x)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -9)))
    e = np.maximum(e, -6.0)                                   # subnormals share the exponent of 2^-6
    q = 2.0 ** (e - 3)
    return np.sign(x) * np.minimum(np.round(a / q) * q, 448.0)


def e4m3_split3(x):
    """The three-term operand split of csrc/attention.hip: split3_e4m3 (x = t0 + t1 / 16 + t2 / 256, residuals in f32)."""
    x = np.asarray(x, np.float32).astype(np.float64)
    t0 = e4m3_round(x)
    r1 = np.float32((x - t0) * 16.0).astype(np.float64)
    t1 = e4m3_round(r1)
    r2 = np.float32((r1 - t1) * 16.0).astype(np.float64)
    t2 = e4m3_round(r2)
    return t0, t1, t2
