"""Whisper encoder/decoder forward in numpy float32 (test oracle; see oracle/__init__.py).

Follows TF/models/whisper/modeling_whisper.py:
  eager attention :215-238, WhisperAttention.forward :284-356 (q scaled by head_dim**-0.5
  before QK^T :309, k_proj has no bias :279, cross K/V computed once :312-335),
  encoder layer :379-413, decoder layer :448-505, encoder :590-646 (conv1+gelu :618,
  conv2(stride 2)+gelu :619, +embed_positions :621-624, final LN :642),
  decoder :688-795 (embed_tokens + learned positions :737-762, final LN :790),
  logits = proj_out(hidden) with proj_out tied to embed_tokens :965, :1080.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
from scipy.special import erf

F32 = np.float32


def layer_norm(x, w, b, eps=1e-5):
    x = x.astype(F32)
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True, dtype=F32)
    return ((x - mu) / np.sqrt(var + F32(eps))) * w + b


def gelu(x):
    """nn.functional.gelu default = exact erf form (TF/activations.py:325)."""
    x = x.astype(F32)
    return (F32(0.5) * x * (F32(1.0) + erf(x * F32(0.7071067811865476)).astype(F32))).astype(F32)


def linear(x, w, b=None):
    y = x @ w.T
    if b is not None:
        y = y + b
    return y.astype(F32)


def conv1d_k3(x, w, b, stride):
    """x [B,C,T], w [O,C,3], padding 1."""
    B, C, T = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1)))
    T_out = (T + 2 - 3) // stride + 1
    cols = np.stack([xp[:, :, k:k + stride * T_out:stride] for k in range(3)], axis=2)  # [B,C,3,T_out]
    y = np.einsum("ock,bckt->bot", w, cols, optimize=True)
    return (y + b[None, :, None]).astype(F32)


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m, dtype=F32)
    return (e / e.sum(axis=axis, keepdims=True, dtype=F32)).astype(F32)


def split_heads(x, n_heads):
    B, T, D = x.shape
    return x.reshape(B, T, n_heads, D // n_heads).transpose(0, 2, 1, 3)


def merge_heads(x):
    B, H, T, Dh = x.shape
    return x.transpose(0, 2, 1, 3).reshape(B, T, H * Dh)


class WhisperOracle:
    def __init__(self, weights: Dict[str, np.ndarray], geom):
        self.w = {k: np.asarray(v, dtype=F32) for k, v in weights.items()}
        self.g = geom
        self.scale = F32((geom.d_model // geom.heads) ** -0.5)

    # ---- encoder -------------------------------------------------------------------
    def encode(self, feats: np.ndarray) -> np.ndarray:
        w, g = self.w, self.g
        x = gelu(conv1d_k3(feats.astype(F32), w["model.encoder.conv1.weight"], w["model.encoder.conv1.bias"], 1))
        x = gelu(conv1d_k3(x, w["model.encoder.conv2.weight"], w["model.encoder.conv2.bias"], 2))
        x = x.transpose(0, 2, 1) + w["model.encoder.embed_positions.weight"][None]
        for i in range(g.enc_layers):
            p = f"model.encoder.layers.{i}"
            h = layer_norm(x, w[p + ".self_attn_layer_norm.weight"], w[p + ".self_attn_layer_norm.bias"])
            q = linear(h, w[p + ".self_attn.q_proj.weight"], w[p + ".self_attn.q_proj.bias"]) * self.scale
            k = linear(h, w[p + ".self_attn.k_proj.weight"])
            v = linear(h, w[p + ".self_attn.v_proj.weight"], w[p + ".self_attn.v_proj.bias"])
            q, k, v = (split_heads(t, g.heads) for t in (q, k, v))
            a = softmax(q @ k.transpose(0, 1, 3, 2)) @ v
            x = x + linear(merge_heads(a), w[p + ".self_attn.out_proj.weight"], w[p + ".self_attn.out_proj.bias"])
            h = layer_norm(x, w[p + ".final_layer_norm.weight"], w[p + ".final_layer_norm.bias"])
            h = gelu(linear(h, w[p + ".fc1.weight"], w[p + ".fc1.bias"]))
            x = x + linear(h, w[p + ".fc2.weight"], w[p + ".fc2.bias"])
        return layer_norm(x, w["model.encoder.layer_norm.weight"], w["model.encoder.layer_norm.bias"]).astype(F32)

    # ---- decoder -------------------------------------------------------------------
    def new_cache(self, enc: np.ndarray):
        """Cross K/V are projected once per generate call (:322-335)."""
        w, g = self.w, self.g
        cache = {"self_k": [None] * g.dec_layers, "self_v": [None] * g.dec_layers, "cross_k": [], "cross_v": [], "len": 0}
        for i in range(g.dec_layers):
            p = f"model.decoder.layers.{i}.encoder_attn"
            cache["cross_k"].append(split_heads(linear(enc, w[p + ".k_proj.weight"]), g.heads))
            cache["cross_v"].append(split_heads(linear(enc, w[p + ".v_proj.weight"], w[p + ".v_proj.bias"]), g.heads))
        return cache

    def decode(self, ids: np.ndarray, cache, want_heads: Optional[List[List[int]]] = None,
               all_logits: bool = False):
        """ids [B, t_new] appended at absolute positions cache['len']..; returns
        (logits of the last position [B,V] f32, cross-attention probs).

        cross-attention: if ``want_heads`` is given -> array [B, H_a, t_new, S] for those
        (layer, head) pairs, else list over layers of [B, H, t_new, S]."""
        w, g = self.w, self.g
        B, t_new = ids.shape
        past = cache["len"]
        x = w["model.decoder.embed_tokens.weight"][ids] + w["model.decoder.embed_positions.weight"][past:past + t_new][None]
        causal = np.triu(np.full((t_new, past + t_new), -np.inf, dtype=F32), k=past + 1)
        cross_probs = []
        for i in range(g.dec_layers):
            p = f"model.decoder.layers.{i}"
            h = layer_norm(x, w[p + ".self_attn_layer_norm.weight"], w[p + ".self_attn_layer_norm.bias"])
            q = split_heads(linear(h, w[p + ".self_attn.q_proj.weight"], w[p + ".self_attn.q_proj.bias"]) * self.scale, g.heads)
            k = split_heads(linear(h, w[p + ".self_attn.k_proj.weight"]), g.heads)
            v = split_heads(linear(h, w[p + ".self_attn.v_proj.weight"], w[p + ".self_attn.v_proj.bias"]), g.heads)
            if cache["self_k"][i] is not None:
                k = np.concatenate([cache["self_k"][i], k], axis=2)
                v = np.concatenate([cache["self_v"][i], v], axis=2)
            cache["self_k"][i], cache["self_v"][i] = k, v
            a = softmax(q @ k.transpose(0, 1, 3, 2) + causal[None, None]) @ v
            x = x + linear(merge_heads(a), w[p + ".self_attn.out_proj.weight"], w[p + ".self_attn.out_proj.bias"])

            h = layer_norm(x, w[p + ".encoder_attn_layer_norm.weight"], w[p + ".encoder_attn_layer_norm.bias"])
            q = split_heads(linear(h, w[p + ".encoder_attn.q_proj.weight"], w[p + ".encoder_attn.q_proj.bias"]) * self.scale, g.heads)
            pr = softmax(q @ cache["cross_k"][i].transpose(0, 1, 3, 2))
            cross_probs.append(pr)
            a = pr @ cache["cross_v"][i]
            x = x + linear(merge_heads(a), w[p + ".encoder_attn.out_proj.weight"], w[p + ".encoder_attn.out_proj.bias"])

            h = layer_norm(x, w[p + ".final_layer_norm.weight"], w[p + ".final_layer_norm.bias"])
            h = gelu(linear(h, w[p + ".fc1.weight"], w[p + ".fc1.bias"]))
            x = x + linear(h, w[p + ".fc2.weight"], w[p + ".fc2.bias"])
        cache["len"] = past + t_new
        x = layer_norm(x, w["model.decoder.layer_norm.weight"], w["model.decoder.layer_norm.bias"])
        hs = x if all_logits else x[:, -1]
        logits = (hs @ w["model.decoder.embed_tokens.weight"].T).astype(F32)
        if want_heads is not None:
            cross = np.stack([cross_probs[l][:, h] for l, h in want_heads], axis=1)  # [B,H_a,t_new,S]
        else:
            cross = cross_probs
        return logits, cross
