"""Weight sources for the engine, keyed by Hugging Face Whisper state-dict names (fp32 numpy arrays).

* ``random_weights``  — seeded synthetic weights of the exact architecture (no checkpoint exists offline;
  bench.py and the parity tests use these and say so).
* ``load_hf_dir``     — a Hugging Face model directory (``model.safetensors`` / sharded), i.e. the artefact
  the reference converts from at whisper_live/backend/faster_whisper_backend.py:133-178.
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict

import numpy as np

from .specs import WhisperSpec


def _sinusoids(length: int, channels: int) -> np.ndarray:
    log_inc = math.log(10000.0) / (channels // 2 - 1)
    inv = np.exp(-log_inc * np.arange(channels // 2, dtype=np.float32)).astype(np.float32)
    scaled = np.arange(length, dtype=np.float32)[:, None] * inv[None, :]
    return np.concatenate([np.sin(scaled), np.cos(scaled)], axis=1).astype(np.float32)


def random_weights(spec: WhisperSpec, seed: int = 0, scale: float = 1.0) -> Dict[str, np.ndarray]:
    """Seeded weights whose activations stay O(1) through the stack (fan-in scaled projections, perturbed
    LayerNorm affine and biases) so numerical parity tests are sensitive in every layer."""
    rng = np.random.default_rng(seed)
    d, F, V = spec.d_model, spec.ffn, spec.vocab
    w: Dict[str, np.ndarray] = {}

    def lin(name, n, k, bias=True, gain=0.7):
        w[name + ".weight"] = (rng.standard_normal((n, k), dtype=np.float32) * (gain * scale / math.sqrt(k))).astype(np.float32)
        if bias:
            w[name + ".bias"] = (rng.standard_normal(n, dtype=np.float32) * 0.05).astype(np.float32)

    def ln(name):
        w[name + ".weight"] = (1.0 + 0.1 * rng.standard_normal(d, dtype=np.float32)).astype(np.float32)
        w[name + ".bias"] = (0.05 * rng.standard_normal(d, dtype=np.float32)).astype(np.float32)

    w["model.encoder.conv1.weight"] = (rng.standard_normal((d, spec.n_mels, 3), dtype=np.float32) * (1.0 / math.sqrt(3 * spec.n_mels))).astype(np.float32)
    w["model.encoder.conv1.bias"] = (rng.standard_normal(d, dtype=np.float32) * 0.05).astype(np.float32)
    w["model.encoder.conv2.weight"] = (rng.standard_normal((d, d, 3), dtype=np.float32) * (1.0 / math.sqrt(3 * d))).astype(np.float32)
    w["model.encoder.conv2.bias"] = (rng.standard_normal(d, dtype=np.float32) * 0.05).astype(np.float32)
    w["model.encoder.embed_positions.weight"] = _sinusoids(spec.n_audio_ctx, d)
    for l in range(spec.enc_layers):
        p = f"model.encoder.layers.{l}."
        ln(p + "self_attn_layer_norm")
        lin(p + "self_attn.q_proj", d, d); lin(p + "self_attn.k_proj", d, d, bias=False); lin(p + "self_attn.v_proj", d, d)
        lin(p + "self_attn.out_proj", d, d, gain=0.5)
        ln(p + "final_layer_norm")
        lin(p + "fc1", F, d); lin(p + "fc2", d, F, gain=0.5)
    ln("model.encoder.layer_norm")
    w["model.decoder.embed_tokens.weight"] = (rng.standard_normal((V, d), dtype=np.float32) * (1.5 / math.sqrt(d))).astype(np.float32)
    w["model.decoder.embed_positions.weight"] = (rng.standard_normal((spec.n_text_ctx, d), dtype=np.float32) * 0.02).astype(np.float32)
    for l in range(spec.dec_layers):
        p = f"model.decoder.layers.{l}."
        ln(p + "self_attn_layer_norm")
        lin(p + "self_attn.q_proj", d, d); lin(p + "self_attn.k_proj", d, d, bias=False); lin(p + "self_attn.v_proj", d, d)
        lin(p + "self_attn.out_proj", d, d, gain=0.5)
        ln(p + "encoder_attn_layer_norm")
        lin(p + "encoder_attn.q_proj", d, d); lin(p + "encoder_attn.k_proj", d, d, bias=False); lin(p + "encoder_attn.v_proj", d, d)
        lin(p + "encoder_attn.out_proj", d, d, gain=0.5)
        ln(p + "final_layer_norm")
        lin(p + "fc1", F, d); lin(p + "fc2", d, F, gain=0.5)
    ln("model.decoder.layer_norm")
    return w


def load_hf_dir(path: str) -> Dict[str, np.ndarray]:
    """Load a Hugging Face Whisper checkpoint directory (safetensors) as fp32 numpy arrays."""
    from safetensors.numpy import load_file

    files = []
    idx = os.path.join(path, "model.safetensors.index.json")
    if os.path.isfile(idx):
        with open(idx) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
    elif os.path.isfile(os.path.join(path, "model.safetensors")):
        files = ["model.safetensors"]
    else:
        raise FileNotFoundError(f"no model.safetensors in {path}")
    sd: Dict[str, np.ndarray] = {}
    for fn in files:
        for k, v in load_file(os.path.join(path, fn)).items():
            if k == "proj_out.weight":
                continue   # tied to model.decoder.embed_tokens.weight
            sd[k if k.startswith("model.") else "model." + k] = np.ascontiguousarray(v.astype(np.float32))
    if "model.encoder.embed_positions.weight" not in sd:
        d = sd["model.encoder.conv1.weight"].shape[0]
        sd["model.encoder.embed_positions.weight"] = _sinusoids(1500, d)
    return sd


# ----------------------------------------------------------------------------------------------------------------
# CTranslate2 model.bin — the artefact the reference actually serves (Systran/faster-whisper-* repositories, or its own
# conversion: whisper_live/backend/faster_whisper_backend.py:133-178). CTranslate2 is not vendored in the reference and
# not installable here, so the container format is restated from its published writer
# (python/ctranslate2/specs/model_spec.py, ModelSpec._serialize, binary version 6):
#   u32 binary_version | str spec_name | u32 spec_revision | u32 n_variables |
#   n x ( str name | u8 rank | rank x u32 dims | u8 dtype | u32 n_bytes | raw little-endian data ) |
#   u32 n_aliases | n x ( str alias | str target )            with  str = u16 length-including-NUL | bytes | NUL
#   dtype ids: 0 float32, 1 int8, 2 int16, 3 int32, 4 float16, 5 bfloat16
# and the variable names from its Whisper spec (specs/whisper_spec.py + converters/transformers.py): fused in_proj
# "linear_0" [3d, d] for self-attention, "linear_0" (query) / "linear_1" (fused key-value [2d, d]) / "linear_2" (output)
# for cross-attention, "gamma"/"beta" for LayerNorm, int8 weights with a per-row float32 "weight_scale"
# (stored value = round(w * scale), scale = 127 / max|row|). UNVERIFIED against a real file in this container (none
# exists offline); tests/test_ct2_loader.py round-trips a file written with the same layout.
_CT2_DTYPES = {0: np.float32, 1: np.int8, 2: np.int16, 3: np.int32, 4: np.float16, 5: None}


def read_ct2_model_bin(path: str):
    """-> (spec_name, revision, {name: ndarray}, {alias: target})"""
    import struct

    with open(path, "rb") as f:
        buf = f.read()
    pos = 0

    def take(fmt):
        nonlocal pos
        v = struct.unpack_from("<" + fmt, buf, pos)
        pos += struct.calcsize("<" + fmt)
        return v[0]

    def take_str():
        nonlocal pos
        n = take("H")
        s = buf[pos: pos + n - 1].decode("utf-8")
        pos += n
        return s

    version = take("I")
    if version < 4 or version > 6:
        raise ValueError(f"{path}: unsupported CTranslate2 binary version {version}")
    spec_name = take_str()
    revision = take("I")
    variables = {}
    for _ in range(take("I")):
        name = take_str()
        rank = take("B")
        shape = [take("I") for _ in range(rank)]
        dtype_id = take("B")
        nbytes = take("I")
        raw = buf[pos: pos + nbytes]
        pos += nbytes
        if dtype_id == 5:       # bfloat16 -> float32 (upper 16 bits)
            arr = (np.frombuffer(raw, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)
        else:
            arr = np.frombuffer(raw, dtype=_CT2_DTYPES[dtype_id])
        variables[name] = arr.reshape(shape) if rank else arr.reshape(())
    aliases = {}
    if pos < len(buf):
        for _ in range(take("I")):
            a = take_str()
            aliases[a] = take_str()
    return spec_name, revision, variables, aliases


def load_ct2_dir(path: str) -> Dict[str, np.ndarray]:
    """A CTranslate2 Whisper model directory (model.bin [+ config.json, tokenizer.json, vocabulary.*]) as the
    Hugging Face state dict the engine consumes (fp32 numpy). int8 / float16 / bfloat16 weights are dequantised; the
    engine repacks everything to fp16 fragments anyway."""
    spec_name, _rev, var, aliases = read_ct2_model_bin(os.path.join(path, "model.bin"))
    if "Whisper" not in spec_name:
        raise ValueError(f"{path}/model.bin holds a {spec_name}, not a Whisper model")

    def get(name):
        v = var[aliases.get(name, name)] if aliases.get(name, name) in var else var[name]
        return v

    def linear_w(prefix):
        w = get(prefix + "/weight")
        if w.dtype == np.int8:
            scale = get(prefix + "/weight_scale").astype(np.float32).reshape(-1, 1)
            return (w.astype(np.float32) / scale).astype(np.float32)
        return np.ascontiguousarray(w.astype(np.float32))

    def f32(name):
        return np.ascontiguousarray(get(name).astype(np.float32))

    sd: Dict[str, np.ndarray] = {}

    def put_ln(dst, src):
        sd[dst + ".weight"], sd[dst + ".bias"] = f32(src + "/gamma"), f32(src + "/beta")

    def put_lin(dst, src, bias=True):
        sd[dst + ".weight"] = linear_w(src)
        if bias:
            sd[dst + ".bias"] = f32(src + "/bias")

    def put_fused(dsts, src, biases):
        w = linear_w(src)
        b = f32(src + "/bias") if (src + "/bias") in var else None
        n = w.shape[0] // len(dsts)
        for i, (dst, has_bias) in enumerate(zip(dsts, biases)):
            sd[dst + ".weight"] = np.ascontiguousarray(w[i * n:(i + 1) * n])
            if has_bias and b is not None:
                sd[dst + ".bias"] = np.ascontiguousarray(b[i * n:(i + 1) * n])

    for c in ("conv1", "conv2"):
        sd[f"model.encoder.{c}.weight"] = linear_w(f"encoder/{c}")
        sd[f"model.encoder.{c}.bias"] = f32(f"encoder/{c}/bias")
    pe = "encoder/position_encodings/encodings"
    d = sd["model.encoder.conv1.weight"].shape[0]
    sd["model.encoder.embed_positions.weight"] = f32(pe) if pe in var else _sinusoids(1500, d)
    put_ln("model.encoder.layer_norm", "encoder/layer_norm")
    l = 0
    while f"encoder/layer_{l}/self_attention/linear_0/weight" in var:
        p, q = f"model.encoder.layers.{l}.", f"encoder/layer_{l}/"
        put_ln(p + "self_attn_layer_norm", q + "self_attention/layer_norm")
        put_fused([p + "self_attn.q_proj", p + "self_attn.k_proj", p + "self_attn.v_proj"], q + "self_attention/linear_0", (True, False, True))
        put_lin(p + "self_attn.out_proj", q + "self_attention/linear_1")
        put_ln(p + "final_layer_norm", q + "ffn/layer_norm")
        put_lin(p + "fc1", q + "ffn/linear_0")
        put_lin(p + "fc2", q + "ffn/linear_1")
        l += 1
    if l == 0:
        raise ValueError("no encoder layers found in model.bin (unexpected variable names)")
    sd["model.decoder.embed_tokens.weight"] = linear_w("decoder/embeddings")
    sd["model.decoder.embed_positions.weight"] = f32("decoder/position_encodings/encodings")
    put_ln("model.decoder.layer_norm", "decoder/layer_norm")
    l = 0
    while f"decoder/layer_{l}/self_attention/linear_0/weight" in var:
        p, q = f"model.decoder.layers.{l}.", f"decoder/layer_{l}/"
        put_ln(p + "self_attn_layer_norm", q + "self_attention/layer_norm")
        put_fused([p + "self_attn.q_proj", p + "self_attn.k_proj", p + "self_attn.v_proj"], q + "self_attention/linear_0", (True, False, True))
        put_lin(p + "self_attn.out_proj", q + "self_attention/linear_1")
        put_ln(p + "encoder_attn_layer_norm", q + "attention/layer_norm")
        put_lin(p + "encoder_attn.q_proj", q + "attention/linear_0")
        put_fused([p + "encoder_attn.k_proj", p + "encoder_attn.v_proj"], q + "attention/linear_1", (False, True))
        put_lin(p + "encoder_attn.out_proj", q + "attention/linear_2")
        put_ln(p + "final_layer_norm", q + "ffn/layer_norm")
        put_lin(p + "fc1", q + "ffn/linear_0")
        put_lin(p + "fc2", q + "ffn/linear_1")
        l += 1
    if l == 0:
        raise ValueError("no decoder layers found in model.bin (unexpected variable names)")
    return sd


def load_model_dir(path: str) -> Dict[str, np.ndarray]:
    """model directory -> state dict: CTranslate2 (model.bin) or Hugging Face (model.safetensors)"""
    if os.path.isfile(os.path.join(path, "model.bin")):
        return load_ct2_dir(path)
    return load_hf_dir(path)
