"""Reader for the CTranslate2 model layout that faster-whisper uses (REF/README.md:186-203 ships CrisperWhisper in that form as
``nyrahealth/faster_CrisperWhisper``): ``model.bin`` + ``config.json`` + ``tokenizer.json`` / ``vocabulary.json``.

PARITY UNPINNED: neither ``ctranslate2`` nor a converted checkpoint is available offline, so this module restates the published
layout -- CTranslate2's ``ModelSpec`` serialisation (binary version 6: ``ctranslate2/specs/model_spec.py``) and the Whisper
variable names its transformers converter emits (``ctranslate2/converters/transformers.py:WhisperLoader``,
``ctranslate2/specs/whisper_spec.py``) -- and is pinned only by a round trip through an independent writer in
``tests/ct2_writer.py``.  Layout::

    uint32  binary_version                     (>= 2 handled)
    string  spec name, uint32 spec revision    (strings: uint16 length including the terminating NUL, then the bytes)
    uint32  n_variables
      string name; uint8 rank; uint32 dims[rank]; uint8 dtype_id (version >= 4, else uint32 item size); uint32 n_bytes; payload
    uint32  n_aliases; (string alias, string target) pairs            (version >= 3)

dtype ids: 0 float32, 1 int8, 2 int16, 3 int32, 4 float16, 5 bfloat16.  int8 / int16 weights come with ``<name>_scale``
(per output row for int8, a scalar for int16): w = q / scale.  Attention projections are fused: self-attention ``linear_0`` =
[q; k; v] (k bias rows are zeros: Whisper's k_proj has none), ``linear_1`` = out; cross-attention ``linear_0`` = q,
``linear_1`` = [k; v], ``linear_2`` = out.  ``decoder/projection/weight`` aliases ``decoder/embeddings/weight``.
"""
from __future__ import annotations

import json
import os
import struct
from typing import Dict, List, Tuple

import numpy as np

_DTYPES = {0: np.float32, 1: np.int8, 2: np.int16, 3: np.int32, 4: np.float16, 5: "bfloat16"}


class _Reader:
    def __init__(self, data: bytes):
        self.d, self.p = data, 0

    def take(self, n: int) -> bytes:
        if n < 0 or self.p + n > len(self.d):
            raise ValueError("truncated CTranslate2 model.bin")
        b = self.d[self.p: self.p + n]
        self.p += n
        return b

    def u(self, fmt: str):
        return struct.unpack("<" + fmt, self.take(struct.calcsize("<" + fmt)))[0]

    def string(self) -> str:
        n = self.u("H")
        return self.take(n).rstrip(b"\0").decode("utf-8")


def read_model_bin(path: str) -> Tuple[str, Dict[str, np.ndarray], Dict[str, str]]:
    """-> (spec name, {variable name: array in its stored dtype (bfloat16 as float32)}, {alias: target})."""
    with open(path, "rb") as f:
        r = _Reader(f.read())
    version = r.u("I")
    if version < 2 or version > 16:
        raise ValueError(f"not a CTranslate2 model.bin (binary version {version})")
    spec_name = r.string()
    r.u("I")                                                   # spec revision
    out: Dict[str, np.ndarray] = {}
    for _ in range(r.u("I")):
        name = r.string()
        rank = r.u("B")
        dims = [r.u("I") for _ in range(rank)]
        if version >= 4:
            dt = _DTYPES.get(r.u("B"))
            n_bytes = r.u("I")
        else:
            item = r.u("B")
            n_bytes = r.u("I") * item
            dt = {4: np.float32, 2: np.int16, 1: np.int8}.get(item)
        if dt is None:
            raise ValueError(f"variable {name}: unknown dtype")
        raw = r.take(n_bytes)
        if dt == "bfloat16":
            a = (np.frombuffer(raw, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)
        else:
            a = np.frombuffer(raw, dtype=dt)
        n_el = int(np.prod(dims)) if dims else 1
        if a.size != n_el:
            raise ValueError(f"variable {name}: {a.size} elements for shape {dims}")
        out[name] = a.reshape(dims) if dims else a.reshape(-1)[0]
    aliases: Dict[str, str] = {}
    if version >= 3 and r.p < len(r.d):
        for _ in range(r.u("I")):
            alias = r.string()
            aliases[alias] = r.string()
    return spec_name, out, aliases


def _dense(v: Dict[str, np.ndarray], name: str) -> np.ndarray:
    """float32 view of a possibly quantised variable."""
    a = v[name]
    if a.dtype == np.int8:
        return a.astype(np.float32) / v[name + "_scale"].astype(np.float32).reshape(-1, *([1] * (a.ndim - 1)))
    if a.dtype == np.int16:
        return a.astype(np.float32) / float(np.asarray(v[name + "_scale"]).reshape(-1)[0])
    return a.astype(np.float32)


def to_hf_state(v: Dict[str, np.ndarray], aliases: Dict[str, str]) -> Dict[str, np.ndarray]:
    """CTranslate2 Whisper variables -> the transformers state_dict names the engine loads (float32)."""
    v = dict(v)
    for alias, target in aliases.items():
        if alias not in v and target in v:
            v[alias] = v[target]
    w: Dict[str, np.ndarray] = {}

    def lin(dst: str, src: str, rows=None, bias=True):
        wt = _dense(v, src + "/weight")
        b = v.get(src + "/bias")
        if rows is not None:
            wt = wt[rows]
            b = None if b is None else b[rows]
        w[dst + ".weight"] = np.ascontiguousarray(wt)
        if bias and b is not None:
            w[dst + ".bias"] = np.ascontiguousarray(b.astype(np.float32))

    def norm(dst: str, src: str):
        w[dst + ".weight"] = v[src + "/gamma"].astype(np.float32)
        w[dst + ".bias"] = v[src + "/beta"].astype(np.float32)

    def attention(dst: str, src: str, d: int, cross: bool):
        if not cross:
            lin(dst + ".q_proj", src + "/linear_0", slice(0, d))
            lin(dst + ".k_proj", src + "/linear_0", slice(d, 2 * d), bias=False)
            lin(dst + ".v_proj", src + "/linear_0", slice(2 * d, 3 * d))
            lin(dst + ".out_proj", src + "/linear_1")
        else:
            lin(dst + ".q_proj", src + "/linear_0")
            lin(dst + ".k_proj", src + "/linear_1", slice(0, d), bias=False)
            lin(dst + ".v_proj", src + "/linear_1", slice(d, 2 * d))
            lin(dst + ".out_proj", src + "/linear_2")

    d = int(v["encoder/conv1/weight"].shape[0])
    for c in ("conv1", "conv2"):
        w[f"model.encoder.{c}.weight"] = _dense(v, f"encoder/{c}/weight")
        w[f"model.encoder.{c}.bias"] = v[f"encoder/{c}/bias"].astype(np.float32)
    w["model.encoder.embed_positions.weight"] = v["encoder/position_encodings/encodings"].astype(np.float32)
    norm("model.encoder.layer_norm", "encoder/layer_norm")
    n_enc = 0
    while f"encoder/layer_{n_enc}/self_attention/linear_0/weight" in v:
        s, t = f"encoder/layer_{n_enc}", f"model.encoder.layers.{n_enc}"
        attention(t + ".self_attn", s + "/self_attention", d, cross=False)
        norm(t + ".self_attn_layer_norm", s + "/self_attention/layer_norm")
        lin(t + ".fc1", s + "/ffn/linear_0"); lin(t + ".fc2", s + "/ffn/linear_1")
        norm(t + ".final_layer_norm", s + "/ffn/layer_norm")
        n_enc += 1
    w["model.decoder.embed_tokens.weight"] = _dense(v, "decoder/embeddings/weight")
    w["model.decoder.embed_positions.weight"] = v["decoder/position_encodings/encodings"].astype(np.float32)
    norm("model.decoder.layer_norm", "decoder/layer_norm")
    n_dec = 0
    while f"decoder/layer_{n_dec}/self_attention/linear_0/weight" in v:
        s, t = f"decoder/layer_{n_dec}", f"model.decoder.layers.{n_dec}"
        attention(t + ".self_attn", s + "/self_attention", d, cross=False)
        norm(t + ".self_attn_layer_norm", s + "/self_attention/layer_norm")
        attention(t + ".encoder_attn", s + "/attention", d, cross=True)
        norm(t + ".encoder_attn_layer_norm", s + "/attention/layer_norm")
        lin(t + ".fc1", s + "/ffn/linear_0"); lin(t + ".fc2", s + "/ffn/linear_1")
        norm(t + ".final_layer_norm", s + "/ffn/layer_norm")
        n_dec += 1
    if n_enc == 0 or n_dec == 0:
        raise ValueError("no Whisper encoder / decoder layers in this CTranslate2 model")
    return w


def geometry(v: Dict[str, np.ndarray]) -> dict:
    """Model geometry from the variable shapes and the scalar ``num_heads`` variables."""
    d, n_mels = int(v["encoder/conv1/weight"].shape[0]), int(v["encoder/conv1/weight"].shape[1])
    enc = sum(1 for k in v if k.startswith("encoder/layer_") and k.endswith("/self_attention/linear_0/weight"))
    dec = sum(1 for k in v if k.startswith("decoder/layer_") and k.endswith("/self_attention/linear_0/weight"))
    heads = v.get("encoder/num_heads", v.get("decoder/num_heads"))
    if heads is None:
        raise ValueError("the CTranslate2 model does not record num_heads")
    return dict(d_model=d, n_mels=n_mels, enc_layers=enc, dec_layers=dec, n_heads=int(np.asarray(heads).reshape(-1)[0]),
                ffn_dim=int(v["encoder/layer_0/ffn/linear_0/weight"].shape[0]),
                vocab_size=int(v["decoder/embeddings/weight"].shape[0]),
                max_target_positions=int(v["decoder/position_encodings/encodings"].shape[0]))


def vocabulary_ids(path: str) -> Dict[str, int]:
    """token string -> id from tokenizer.json (preferred) or CTranslate2's vocabulary.json (a list, index = id)."""
    tj = os.path.join(path, "tokenizer.json")
    if os.path.exists(tj):
        j = json.load(open(tj, encoding="utf-8"))
        ids = {s: int(i) for s, i in j["model"]["vocab"].items()}
        for a in j.get("added_tokens", []):
            ids[a["content"]] = int(a["id"])
        return ids
    for name in ("vocabulary.json", "vocabulary.txt"):
        p = os.path.join(path, name)
        if os.path.exists(p):
            toks: List[str] = json.load(open(p, encoding="utf-8")) if name.endswith(".json") else open(p, encoding="utf-8").read().split("\n")
            return {s: i for i, s in enumerate(toks)}
    raise FileNotFoundError(f"neither tokenizer.json nor vocabulary.json under {path}")
