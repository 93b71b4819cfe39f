"""CPU oracle for the Whisper network (torch fp32 on CPU). Test infrastructure only — see oracle/__init__.py.

Restates what ctranslate2.models.Whisper.encode / the decoder inside .generate compute for the reference
(whisper_live/transcriber/transcriber_faster_whisper.py:1339-1348, :1394-1407). CTranslate2 is un-vendored; the
network definition followed here is transformers models/whisper/modeling_whisper.py (identical maths):
conv stem :566-567/:618-619, fixed sinusoids :55, pre-LN encoder layer :360-402, decoder layer
(self -> cross -> MLP) :416-498, q scaling head_dim**-0.5 on q :267/:309, k_proj without bias :278,
learned decoder positions :204-210, final LayerNorms :642/:790, tied output projection :965-970.
Weights use the Hugging Face state-dict names.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class Spec:
    n_mels: int
    d_model: int
    n_heads: int
    enc_layers: int
    dec_layers: int
    ffn: int
    vocab: int
    n_audio_ctx: int = 1500
    n_text_ctx: int = 448


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    """Whisper's fixed encoder position table (modeling_whisper.py:55)."""
    log_inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-log_inc * torch.arange(channels // 2, dtype=torch.float32))
    scaled = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([scaled.sin(), scaled.cos()], dim=1)


def quantize_rows_int8(w: torch.Tensor):
    """Per-output-row symmetric int8: q[n, :] = round(w[n, :] * s[n]), s[n] = 127 / max|w[n, :]| — the weight scheme CTranslate2's
    converter applies for compute_type int8 (the reference's CPU default, whisper_live/backend/faster_whisper_backend.py:88-93; CT2 is
    un-vendored: restated from its published quantisation description). Returns (integer-valued float32 [N, K], scale [N])."""
    amax = w.abs().amax(dim=1).clamp_min(1e-12)
    s = 127.0 / amax
    return torch.round(w * s[:, None]).clamp_(-127, 127), s


class WhisperOracle:
    """int8 = None   fp32 arithmetic (the parity oracle of the HIP path);
       int8 = "ct2"  every linear layer (and the tied output projection) in CTranslate2's CPU int8 arithmetic: weights per-output-row
                     symmetric int8, activations quantised dynamically per row (s = 127 / max|x_row|, round to nearest), integer
                     accumulation, one dequantisation per output — convolutions, LayerNorm, softmax, GELU and the embedding gather stay
                     float. Used to BOUND the distance between the HIP fp16 path and the arithmetic of the reference's CPU backend
                     (tests/test_int8_gap.py);
       int8 = "fbgemm" the same layers as torch's dynamic-quantised Linear (per-tensor activation scale): bench.py's `port-int8`
                     timing leg only."""

    def __init__(self, spec: Spec, weights: Dict[str, np.ndarray], dtype=torch.float32, int8: Optional[str] = None):
        self.spec = spec
        self.w = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in weights.items()}
        self.dtype = dtype
        self.H = spec.n_heads
        self.int8 = int8
        self.q: Dict[str, tuple] = {}
        self.qmod: Dict[str, object] = {}
        if int8 not in (None, "ct2", "fbgemm"):
            raise ValueError(f"int8 mode {int8!r}")
        if int8:
            lin = [k[: -len(".weight")] for k, v in self.w.items()
                   if k.endswith(".weight") and v.ndim == 2 and "embed_positions" not in k and "layer_norm" not in k]
            for pre in lin:
                W = self.w[pre + ".weight"].float()
                if int8 == "ct2":
                    self.q[pre] = quantize_rows_int8(W)
                else:
                    import warnings
                    import torch.nn as nn
                    m = nn.Linear(W.shape[1], W.shape[0], bias=True)
                    with torch.no_grad():
                        m.weight.copy_(W)
                        b = self.w.get(pre + ".bias")
                        m.bias.copy_(b.float() if b is not None else torch.zeros(W.shape[0]))
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        self.qmod[pre] = torch.ao.quantization.quantize_dynamic(nn.Sequential(m), {nn.Linear}, dtype=torch.qint8)[0]

    # ------------------------------------------------------------------ helpers
    def _ln(self, x, prefix):
        return F.layer_norm(x, (x.shape[-1],), self.w[prefix + ".weight"], self.w[prefix + ".bias"], 1e-5)

    def _lin(self, x, prefix, bias=True):
        b = self.w.get(prefix + ".bias") if bias else None
        if self.int8 == "ct2":
            Wq, ws = self.q[prefix]
            xs = 127.0 / x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-12)
            xq = torch.round(x * xs).clamp_(-127, 127)
            y = (xq @ Wq.T) / (xs * ws)              # integer-valued operands: the float product IS the int32 accumulation (|sum| << 2^31)
            return y if b is None else y + b
        if self.int8 == "fbgemm":
            return self.qmod[prefix](x)              # (k_proj: its module carries a zero bias)
        return F.linear(x, self.w[prefix + ".weight"], b)

    def _logits(self, x):
        """tied output projection h . tok_emb^T (modeling_whisper.py:965-970)"""
        if self.int8:
            return self._lin(x, "model.decoder.embed_tokens", bias=False)
        return x @ self.w["model.decoder.embed_tokens.weight"].T

    def _heads(self, x):  # [B, T, d] -> [B, H, T, 64]
        B, T, d = x.shape
        return x.view(B, T, self.H, d // self.H).transpose(1, 2)

    def _attend(self, q, k, v, mask=None):
        s = q @ k.transpose(-1, -2)
        if mask is not None:
            s = s + mask
        p = torch.softmax(s, dim=-1)
        o = p @ v
        B, H, T, hd = o.shape
        return o.transpose(1, 2).reshape(B, T, H * hd)

    # ------------------------------------------------------------------ encoder
    @torch.no_grad()
    def encode(self, features: np.ndarray) -> torch.Tensor:
        """features float32 [B, n_mels, 3000] -> [B, 1500, d]."""
        x = torch.as_tensor(np.ascontiguousarray(features)).to(self.dtype)
        x = F.gelu(F.conv1d(x, self.w["model.encoder.conv1.weight"], self.w["model.encoder.conv1.bias"], padding=1))
        x = F.gelu(F.conv1d(x, self.w["model.encoder.conv2.weight"], self.w["model.encoder.conv2.bias"], stride=2, padding=1))
        x = x.permute(0, 2, 1) + self.w["model.encoder.embed_positions.weight"]
        scale = (self.spec.d_model // self.H) ** -0.5
        for l in range(self.spec.enc_layers):
            p = f"model.encoder.layers.{l}."
            h = self._ln(x, p + "self_attn_layer_norm")
            q = self._heads(self._lin(h, p + "self_attn.q_proj") * scale)
            k = self._heads(self._lin(h, p + "self_attn.k_proj", bias=False))
            v = self._heads(self._lin(h, p + "self_attn.v_proj"))
            x = x + self._lin(self._attend(q, k, v), p + "self_attn.out_proj")
            h = self._ln(x, p + "final_layer_norm")
            x = x + self._lin(F.gelu(self._lin(h, p + "fc1")), p + "fc2")
        return self._ln(x, "model.encoder.layer_norm")

    # ------------------------------------------------------------------ decoder
    @torch.no_grad()
    def cross_kv(self, enc: torch.Tensor):
        out = []
        for l in range(self.spec.dec_layers):
            p = f"model.decoder.layers.{l}.encoder_attn."
            out.append((self._heads(self._lin(enc, p + "k_proj", bias=False)), self._heads(self._lin(enc, p + "v_proj"))))
        return out

    @torch.no_grad()
    def decode_logits(self, enc: torch.Tensor, tokens: np.ndarray, ckv=None) -> torch.Tensor:
        """Teacher-forced decoder: enc [B,1500,d], tokens int [B,T] -> logits [B,T,V] (causal)."""
        tok = torch.as_tensor(np.asarray(tokens)).long()
        B, T = tok.shape
        ckv = ckv or self.cross_kv(enc)
        x = self.w["model.decoder.embed_tokens.weight"][tok] + self.w["model.decoder.embed_positions.weight"][:T]
        scale = (self.spec.d_model // self.H) ** -0.5
        mask = torch.full((T, T), float("-inf")).triu(1)
        for l in range(self.spec.dec_layers):
            p = f"model.decoder.layers.{l}."
            h = self._ln(x, p + "self_attn_layer_norm")
            q = self._heads(self._lin(h, p + "self_attn.q_proj") * scale)
            k = self._heads(self._lin(h, p + "self_attn.k_proj", bias=False))
            v = self._heads(self._lin(h, p + "self_attn.v_proj"))
            x = x + self._lin(self._attend(q, k, v, mask), p + "self_attn.out_proj")
            h = self._ln(x, p + "encoder_attn_layer_norm")
            q = self._heads(self._lin(h, p + "encoder_attn.q_proj") * scale)
            x = x + self._lin(self._attend(q, ckv[l][0], ckv[l][1]), p + "encoder_attn.out_proj")
            h = self._ln(x, p + "final_layer_norm")
            x = x + self._lin(F.gelu(self._lin(h, p + "fc1")), p + "fc2")
        x = self._ln(x, "model.decoder.layer_norm")
        return self._logits(x)


class StepDecoder:
    """Incremental decoder with self-attention KV cache for the search loop (rows = hypotheses)."""

    def __init__(self, model: WhisperOracle, enc: torch.Tensor):
        self.m = model
        self.enc = enc                      # [1, 1500, d] (one audio item)
        self.ckv = model.cross_kv(enc)
        self.k: List[Optional[torch.Tensor]] = [None] * model.spec.dec_layers   # [rows, H, t, 64]
        self.v: List[Optional[torch.Tensor]] = [None] * model.spec.dec_layers
        self.t = 0

    def reorder(self, parents: List[int]):
        idx = torch.as_tensor(parents).long()
        self.k = [None if k is None else k[idx] for k in self.k]
        self.v = [None if v is None else v[idx] for v in self.v]

    @torch.no_grad()
    def step(self, tokens: np.ndarray) -> torch.Tensor:
        """tokens int [rows, n] fed at positions t..t+n-1 -> logits [rows, n, V]."""
        m = self.m
        tok = torch.as_tensor(np.asarray(tokens)).long()
        rows, n = tok.shape
        x = m.w["model.decoder.embed_tokens.weight"][tok] + m.w["model.decoder.embed_positions.weight"][self.t : self.t + n]
        scale = (m.spec.d_model // m.H) ** -0.5
        total = self.t + n
        mask = torch.full((n, total), float("-inf")).triu(self.t + 1)
        for l in range(m.spec.dec_layers):
            p = f"model.decoder.layers.{l}."
            h = m._ln(x, p + "self_attn_layer_norm")
            q = m._heads(m._lin(h, p + "self_attn.q_proj") * scale)
            k = m._heads(m._lin(h, p + "self_attn.k_proj", bias=False))
            v = m._heads(m._lin(h, p + "self_attn.v_proj"))
            if self.k[l] is not None:
                kp, vp = self.k[l], self.v[l]
                if kp.shape[0] != rows:
                    kp, vp = kp.expand(rows, -1, -1, -1), vp.expand(rows, -1, -1, -1)
                k, v = torch.cat([kp, k], dim=2), torch.cat([vp, v], dim=2)
            self.k[l], self.v[l] = k, v
            x = x + m._lin(m._attend(q, k, v, mask), p + "self_attn.out_proj")
            h = m._ln(x, p + "encoder_attn_layer_norm")
            q = m._heads(m._lin(h, p + "encoder_attn.q_proj") * scale)
            x = x + m._lin(m._attend(q, self.ckv[l][0], self.ckv[l][1]), p + "encoder_attn.out_proj")
            h = m._ln(x, p + "final_layer_norm")
            x = x + m._lin(F.gelu(m._lin(h, p + "fc1")), p + "fc2")
        self.t = total
        x = m._ln(x, "model.decoder.layer_norm")
        return m._logits(x)
