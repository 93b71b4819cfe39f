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
