"""Independent writer of the CTranslate2 model layout (binary version 6) for the round-trip test of crisperwhisper_amd/ct2.py:
takes transformers-named Whisper weights and emits ``model.bin`` the way CTranslate2's converter lays a WhisperSpec out
(fused attention projections with a zero k bias, alias for the tied projection, scalar bookkeeping variables), optionally
in float16 or with int8 weight quantisation (per-row scale = 127 / max|w|, w_q = round(w * scale))."""
import struct

import numpy as np

_IDS = {np.dtype(np.float32): 0, np.dtype(np.int8): 1, np.dtype(np.int16): 2, np.dtype(np.int32): 3, np.dtype(np.float16): 4}


def _s(x: str) -> bytes:
    b = x.encode("utf-8") + b"\0"
    return struct.pack("<H", len(b)) + b


def write_model_bin(path, weights, n_heads, enc_layers, dec_layers, storage="float32"):
    var = {}

    def put(name, a, quantisable=False):
        a = np.asarray(a)
        if quantisable and storage == "int8" and a.ndim == 2:
            amax = np.abs(a).max(axis=1)
            scale = (127.0 / np.where(amax == 0, 127.0, amax)).astype(np.float32)
            var[name] = np.clip(np.round(a * scale[:, None]), -127, 127).astype(np.int8)
            var[name + "_scale"] = scale
        elif storage == "float16" and a.dtype == np.float32 and a.ndim >= 1:
            var[name] = a.astype(np.float16)
        else:
            var[name] = a

    def lin(dst, src_list):
        ws = [weights[s + ".weight"] for s in src_list]
        bs = [weights.get(s + ".bias", np.zeros(weights[s + ".weight"].shape[0], np.float32)) for s in src_list]
        put(dst + "/weight", np.concatenate(ws, 0), quantisable=True)
        put(dst + "/bias", np.concatenate(bs, 0))

    def norm(dst, src):
        put(dst + "/gamma", weights[src + ".weight"]); put(dst + "/beta", weights[src + ".bias"])

    for c in ("conv1", "conv2"):
        put(f"encoder/{c}/weight", weights[f"model.encoder.{c}.weight"]); put(f"encoder/{c}/bias", weights[f"model.encoder.{c}.bias"])
    put("encoder/position_encodings/encodings", weights["model.encoder.embed_positions.weight"])
    norm("encoder/layer_norm", "model.encoder.layer_norm")
    var["encoder/num_heads"] = np.array(n_heads, np.int16)
    var["encoder/pre_norm"] = np.array(1, np.int8)
    for l in range(enc_layers):
        s, t = f"model.encoder.layers.{l}", f"encoder/layer_{l}"
        lin(t + "/self_attention/linear_0", [s + ".self_attn.q_proj", s + ".self_attn.k_proj", s + ".self_attn.v_proj"])
        lin(t + "/self_attention/linear_1", [s + ".self_attn.out_proj"])
        norm(t + "/self_attention/layer_norm", s + ".self_attn_layer_norm")
        lin(t + "/ffn/linear_0", [s + ".fc1"]); lin(t + "/ffn/linear_1", [s + ".fc2"])
        norm(t + "/ffn/layer_norm", s + ".final_layer_norm")
    put("decoder/embeddings/weight", weights["model.decoder.embed_tokens.weight"], quantisable=True)
    put("decoder/position_encodings/encodings", weights["model.decoder.embed_positions.weight"])
    norm("decoder/layer_norm", "model.decoder.layer_norm")
    var["decoder/num_heads"] = np.array(n_heads, np.int16)
    var["decoder/scale_embeddings"] = np.array(0, np.int8)
    for l in range(dec_layers):
        s, t = f"model.decoder.layers.{l}", f"decoder/layer_{l}"
        lin(t + "/self_attention/linear_0", [s + ".self_attn.q_proj", s + ".self_attn.k_proj", s + ".self_attn.v_proj"])
        lin(t + "/self_attention/linear_1", [s + ".self_attn.out_proj"])
        norm(t + "/self_attention/layer_norm", s + ".self_attn_layer_norm")
        lin(t + "/attention/linear_0", [s + ".encoder_attn.q_proj"])
        lin(t + "/attention/linear_1", [s + ".encoder_attn.k_proj", s + ".encoder_attn.v_proj"])
        lin(t + "/attention/linear_2", [s + ".encoder_attn.out_proj"])
        norm(t + "/attention/layer_norm", s + ".encoder_attn_layer_norm")
        lin(t + "/ffn/linear_0", [s + ".fc1"]); lin(t + "/ffn/linear_1", [s + ".fc2"])
        norm(t + "/ffn/layer_norm", s + ".final_layer_norm")
    out = struct.pack("<I", 6) + _s("WhisperSpec") + struct.pack("<I", 3) + struct.pack("<I", len(var))
    for name in sorted(var):
        a = np.ascontiguousarray(var[name])
        out += _s(name) + struct.pack("<B", a.ndim) + b"".join(struct.pack("<I", d) for d in a.shape)
        out += struct.pack("<B", _IDS[a.dtype]) + struct.pack("<I", a.nbytes) + a.tobytes()
    out += struct.pack("<I", 1) + _s("decoder/projection/weight") + _s("decoder/embeddings/weight")
    with open(path, "wb") as f:
        f.write(out)
