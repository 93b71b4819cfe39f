"""CPU test of the CTranslate2 model.bin reader (whisperlive_amd/weights.py): a file written with the published
container layout and CTranslate2's Whisper variable names (fused in_proj, fused key-value, gamma/beta, int8 +
weight_scale, an alias for the tied projection) must come back as the Hugging Face state dict it was made from.
The layout itself is restated, not verified against a real CTranslate2 file (none exists offline) — see weights.py."""
import os
import struct

import numpy as np
import pytest

from whisperlive_amd.specs import WhisperSpec
from whisperlive_amd.weights import load_ct2_dir, load_model_dir, random_weights, read_ct2_model_bin

DT = {np.dtype(np.float32): 0, np.dtype(np.int8): 1, np.dtype(np.int16): 2, np.dtype(np.int32): 3, np.dtype(np.float16): 4}


def _wstr(f, s):
    b = s.encode("utf-8")
    f.write(struct.pack("<H", len(b) + 1)); f.write(b); f.write(b"\0")


def _write_ct2(path, variables, aliases):
    with open(path, "wb") as f:
        f.write(struct.pack("<I", 6)); _wstr(f, "WhisperSpec"); f.write(struct.pack("<I", 3))
        f.write(struct.pack("<I", len(variables)))
        for name, v in variables.items():
            v = np.ascontiguousarray(v)
            _wstr(f, name)
            f.write(struct.pack("<B", v.ndim))
            for dim in v.shape:
                f.write(struct.pack("<I", dim))
            f.write(struct.pack("<B", DT[v.dtype])); f.write(struct.pack("<I", v.nbytes)); f.write(v.tobytes())
        f.write(struct.pack("<I", len(aliases)))
        for a, t in aliases.items():
            _wstr(f, a); _wstr(f, t)


def _to_ct2(sd, spec, quant):
    """HF state dict -> CTranslate2 variables (what converters/transformers.py WhisperLoader produces)"""
    var = {}

    def lin(dst, w, b=None):
        if quant == "int8":
            scale = (127.0 / np.abs(w).max(axis=1)).astype(np.float32)
            var[dst + "/weight"] = np.round(w * scale[:, None]).astype(np.int8)
            var[dst + "/weight_scale"] = scale
        else:
            var[dst + "/weight"] = w.astype(np.float16 if quant == "float16" else np.float32)
        if b is not None:
            var[dst + "/bias"] = b.astype(np.float32)

    def ln(dst, src):
        var[dst + "/gamma"], var[dst + "/beta"] = sd[src + ".weight"], sd[src + ".bias"]

    d = spec.d_model
    for c in ("conv1", "conv2"):
        var[f"encoder/{c}/weight"] = sd[f"model.encoder.{c}.weight"].astype(np.float16 if quant != "float32" else np.float32)
        var[f"encoder/{c}/bias"] = sd[f"model.encoder.{c}.bias"]
    var["encoder/position_encodings/encodings"] = sd["model.encoder.embed_positions.weight"]
    ln("encoder/layer_norm", "model.encoder.layer_norm")
    zero = np.zeros(d, np.float32)
    for side, n in (("encoder", spec.enc_layers), ("decoder", spec.dec_layers)):
        for l in range(n):
            p, q = f"model.{side}.layers.{l}.", f"{side}/layer_{l}/"
            ln(q + "self_attention/layer_norm", p + "self_attn_layer_norm")
            lin(q + "self_attention/linear_0",
                np.concatenate([sd[p + f"self_attn.{x}_proj.weight"] for x in "qkv"]),
                np.concatenate([sd[p + "self_attn.q_proj.bias"], zero, sd[p + "self_attn.v_proj.bias"]]))
            lin(q + "self_attention/linear_1", sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
            if side == "decoder":
                ln(q + "attention/layer_norm", p + "encoder_attn_layer_norm")
                lin(q + "attention/linear_0", sd[p + "encoder_attn.q_proj.weight"], sd[p + "encoder_attn.q_proj.bias"])
                lin(q + "attention/linear_1", np.concatenate([sd[p + "encoder_attn.k_proj.weight"], sd[p + "encoder_attn.v_proj.weight"]]),
                    np.concatenate([zero, sd[p + "encoder_attn.v_proj.bias"]]))
                lin(q + "attention/linear_2", sd[p + "encoder_attn.out_proj.weight"], sd[p + "encoder_attn.out_proj.bias"])
            ln(q + "ffn/layer_norm", p + "final_layer_norm")
            lin(q + "ffn/linear_0", sd[p + "fc1.weight"], sd[p + "fc1.bias"])
            lin(q + "ffn/linear_1", sd[p + "fc2.weight"], sd[p + "fc2.bias"])
    lin("decoder/embeddings", sd["model.decoder.embed_tokens.weight"])
    var["decoder/position_encodings/encodings"] = sd["model.decoder.embed_positions.weight"]
    ln("decoder/layer_norm", "model.decoder.layer_norm")
    var["decoder/scale_embeddings"] = np.asarray(False).astype(np.int8)
    return var, {"decoder/projection/weight": "decoder/embeddings/weight"}


@pytest.mark.parametrize("quant", ["float32", "float16", "int8"])
def test_ct2_model_bin_round_trip(tmp_path, quant):
    spec = WhisperSpec(n_mels=80, d_model=128, n_heads=2, enc_layers=2, dec_layers=3, ffn=512, vocab=2310)
    sd = random_weights(spec, seed=5)
    var, aliases = _to_ct2(sd, spec, quant)
    _write_ct2(os.path.join(tmp_path, "model.bin"), var, aliases)
    name, rev, rv, ra = read_ct2_model_bin(os.path.join(tmp_path, "model.bin"))
    assert name == "WhisperSpec" and rev == 3 and set(rv) == set(var) and ra == aliases
    got = load_model_dir(str(tmp_path))                     # model.bin present -> CTranslate2 path
    assert got.keys() == load_ct2_dir(str(tmp_path)).keys()
    missing = {k for k in sd if k not in got and not k.endswith("k_proj.bias")}
    assert not missing, missing
    for k, ref in sd.items():
        if k not in got:
            continue
        tol = dict(float32=0.0, float16=2e-3, int8=2e-2)[quant] * (np.abs(ref).max() + 1e-9)
        if "embed_positions" in k or ".bias" in k or "layer_norm" in k:
            tol = 0.0 if quant == "float32" or "conv" not in k else tol
        assert got[k].dtype == np.float32 and got[k].shape == ref.shape, k
        assert np.abs(got[k] - ref).max() <= tol + 1e-12, (k, float(np.abs(got[k] - ref).max()), tol)
    # k_proj has no bias in Whisper: CTranslate2 stores zeros in the fused bias, the loader must not invent one
    assert not any(k.endswith("k_proj.bias") for k in got)
