"""oracle/alignment.py — CPU restatement of ctranslate2.models.Whisper.align as the reference calls it
(whisper_live/transcriber/transcriber_faster_whisper.py:1657-1663). TEST INFRASTRUCTURE ONLY.

CTranslate2's source is not in the reference tree; the algorithm is the published one it reimplements —
openai/whisper timing.py ``find_alignment`` / ``median_filter`` / ``dtw_cpu``, identical to Hugging Face
transformers models/whisper/generation_whisper.py ``_median_filter`` (:64) and ``_dynamic_time_warping`` (:43),
against which ``median_filter`` and ``dtw`` below are pinned (tests/golden/align_golden.npz, made by
tests/golden/make_align_golden.py). The attention / logits half is pinned through oracle/model.py (HF encoder states
and logits fixtures)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

from .model import WhisperOracle


def median_filter(x: np.ndarray, width: int) -> np.ndarray:
    """median over a sliding window of odd `width` along the last axis, reflect padding; returned unchanged when the
    axis is not longer than the padding (as the reference implementation does)."""
    pad = width // 2
    if width <= 1 or x.shape[-1] <= pad:
        return x
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(pad, pad)], mode="reflect")
    win = np.lib.stride_tricks.sliding_window_view(xp, width, axis=-1)
    return np.sort(win, axis=-1)[..., pad]


def dtw(x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """dynamic time warping over a cost matrix [N, M] -> (text_indices, time_indices) of the monotone minimal path."""
    N, M = x.shape
    cost = np.full((N + 1, M + 1), np.inf, dtype=np.float32)
    trace = -np.ones((N + 1, M + 1), dtype=np.int64)
    cost[0, 0] = 0
    for j in range(1, M + 1):
        for i in range(1, N + 1):
            c0, c1, c2 = cost[i - 1, j - 1], cost[i - 1, j], cost[i, j - 1]
            if c0 < c1 and c0 < c2:
                c, t = c0, 0
            elif c1 < c0 and c1 < c2:
                c, t = c1, 1
            else:
                c, t = c2, 2
            cost[i, j] = x[i - 1, j - 1] + c
            trace[i, j] = t
    i, j = N, M
    trace[0, :] = 2
    trace[:, 0] = 1
    ti, fi = [], []
    while i > 0 or j > 0:
        ti.append(i - 1); fi.append(j - 1)
        if trace[i, j] == 0:
            i -= 1; j -= 1
        elif trace[i, j] == 1:
            i -= 1
        else:
            j -= 1
    return np.asarray(ti[::-1]), np.asarray(fi[::-1])


@torch.no_grad()
def cross_qk(model: WhisperOracle, enc: torch.Tensor, tokens: Sequence[int]) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """teacher-forced pass: logits [T, V] and, per decoder layer, the scaled cross-attention scores [H, T, 1500]."""
    import torch.nn.functional as F
    m = model
    tok = torch.as_tensor(np.asarray(tokens)).long()[None]
    T = tok.shape[1]
    ckv = m.cross_kv(enc)
    x = m.w["model.decoder.embed_tokens.weight"][tok] + m.w["model.decoder.embed_positions.weight"][:T]
    scale = (m.spec.d_model // m.H) ** -0.5
    mask = torch.full((T, T), float("-inf")).triu(1)
    qks = []
    for l in range(m.spec.dec_layers):
        p = f"model.decoder.layers.{l}."
        h = m._ln(x, p + "self_attn_layer_norm")
        q = m._heads(m._lin(h, p + "self_attn.q_proj") * scale)
        k = m._heads(m._lin(h, p + "self_attn.k_proj", bias=False))
        v = m._heads(m._lin(h, p + "self_attn.v_proj"))
        x = x + m._lin(m._attend(q, k, v, mask), p + "self_attn.out_proj")
        h = m._ln(x, p + "encoder_attn_layer_norm")
        q = m._heads(m._lin(h, p + "encoder_attn.q_proj") * scale)
        qks.append((q @ ckv[l][0].transpose(-1, -2))[0])            # [H, T, 1500]
        x = x + m._lin(m._attend(q, ckv[l][0], ckv[l][1]), p + "encoder_attn.out_proj")
        h = m._ln(x, p + "final_layer_norm")
        x = x + m._lin(F.gelu(m._lin(h, p + "fc1")), p + "fc2")
    x = m._ln(x, "model.decoder.layer_norm")
    return (x @ m.w["model.decoder.embed_tokens.weight"].T)[0], qks


def align(model: WhisperOracle, enc: torch.Tensor, sot_sequence: Sequence[int], no_timestamps: int, text_tokens: Sequence[int],
          eot: int, num_frames: int, heads: Sequence[Tuple[int, int]], median_filter_width: int = 7):
    """-> (text_indices, time_indices, text_token_probs, matrix) for ONE item (enc [1, 1500, d])."""
    tokens = list(sot_sequence) + [no_timestamps] + list(text_tokens) + [eot]
    n_sot = len(sot_sequence)
    logits, qks = cross_qk(model, enc, tokens)
    sampled = logits[n_sot:, :eot]
    probs = sampled.softmax(dim=-1)
    text_token_probs = probs[np.arange(len(text_tokens)), list(text_tokens)].numpy()
    nf = max(1, min(1500, num_frames // 2))
    w = torch.stack([qks[l][h] for l, h in heads])[:, :, :nf].softmax(dim=-1)          # [heads, T, nf]
    std, mean = torch.std_mean(w, dim=-2, keepdim=True, unbiased=False)
    w = ((w - mean) / std).numpy()
    w = median_filter(w, median_filter_width)
    matrix = w.mean(axis=0)[n_sot:-1]
    ti, fi = dtw(-matrix)
    return ti, fi, text_token_probs, matrix
