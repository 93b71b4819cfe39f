"""CPU restatement of the Silero-VAD speech-probability network (16 kHz branch) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(whisperlive_amd.vad.SileroHIPModel -> libwlx.so `wlx_vad_*`) never does.

PARITY UNPINNED. The network is the one `faster_whisper.vad` runs through onnxruntime on the reference's path
(whisper_live/transcriber/transcriber_faster_whisper.py:830-838; faster-whisper==1.2.0, requirements/server.txt:1 —
un-vendored, and neither it, onnxruntime nor the Silero weights exist offline). The I/O contract is the one the
reference documents in-tree for the same model (whisper_live/vad.py:50-109): 512-sample windows at 16 kHz, each
prefixed with the last 64 samples of the previous window (zeros before the first), recurrent state [2, B, 128], one
probability per window. The layer stack below is restated from the published Silero-VAD v5 model description
(learned-STFT conv n_fft 256 / hop 128 -> magnitude -> 4 x (conv1d k3 + ReLU) -> LSTM cell 128
-> ReLU -> conv1d 128->1 -> sigmoid). Until a real weight file can be run through both, what the tests pin is
(i) this restatement against an independent torch.nn.functional build of the same stack, and (ii) the HIP kernels
against this restatement.

Weights (all float32), names used across oracle / loader / engine:
  stft_basis  [258, 256]   rows 0..128 = real part filters, 129..257 = imaginary part filters (Fourier basis x Hann)
  enc{0..3}_w [Cout, Cin, 3], enc{0..3}_b [Cout]   channels 129->128 (stride 1), 128->64 (2), 64->64 (2), 64->128 (1); pad 1
  lstm_w_ih   [512, 128], lstm_w_hh [512, 128], lstm_b_ih [512], lstm_b_hh [512]   gate order i, f, g, o
  out_w       [128], out_b [1]
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

WINDOW = 512
CONTEXT = 64
N_FFT = 256
HOP = 128
N_BINS = N_FFT // 2 + 1            # 129
HIDDEN = 128
ENC_CHANNELS = ((129, 128, 1), (128, 64, 2), (64, 64, 2), (64, 128, 1))     # (Cin, Cout, stride)


def fourier_basis() -> np.ndarray:
    """The fixed STFT filter bank of the model: Hann-windowed cos / -sin rows, real part first."""
    n = np.arange(N_FFT)
    k = np.arange(N_BINS)[:, None]
    win = 0.5 - 0.5 * np.cos(2 * np.pi * n / N_FFT)           # periodic Hann
    ang = 2 * np.pi * k * n / N_FFT
    return np.concatenate([np.cos(ang) * win, -np.sin(ang) * win], axis=0).astype(np.float32)


def random_weights(seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded stand-in weights of the exact shapes (fan-in scaled so activations stay O(1) and probabilities spread
    over (0, 1) instead of saturating)."""
    rng = np.random.default_rng(seed)
    w = {"stft_basis": fourier_basis()}
    for i, (cin, cout, _s) in enumerate(ENC_CHANNELS):
        w[f"enc{i}_w"] = (rng.standard_normal((cout, cin, 3)) * (1.6 / np.sqrt(3 * cin))).astype(np.float32)
        w[f"enc{i}_b"] = (rng.standard_normal(cout) * 0.05).astype(np.float32)
    s = 1.0 / np.sqrt(HIDDEN)
    w["lstm_w_ih"] = (rng.uniform(-s, s, (4 * HIDDEN, HIDDEN)) * 2.0).astype(np.float32)
    w["lstm_w_hh"] = (rng.uniform(-s, s, (4 * HIDDEN, HIDDEN)) * 2.0).astype(np.float32)
    w["lstm_b_ih"] = rng.uniform(-s, s, 4 * HIDDEN).astype(np.float32)
    w["lstm_b_hh"] = rng.uniform(-s, s, 4 * HIDDEN).astype(np.float32)
    w["out_w"] = (rng.standard_normal(HIDDEN) * 0.8).astype(np.float32)
    w["out_b"] = np.asarray([0.0], np.float32)
    return w


def frame_windows(audio: np.ndarray) -> np.ndarray:
    """f32[n] -> f32[ceil(n/512), 576]: zero-pad to a multiple of 512, prefix every window with the last 64 samples of
    the previous one (zeros for the first) — whisper_live/vad.py:73-86,97-104."""
    audio = np.asarray(audio, np.float32).reshape(-1)
    pad = (-audio.shape[0]) % WINDOW
    x = np.pad(audio, (0, pad)).reshape(-1, WINDOW)
    ctx = np.zeros((x.shape[0], CONTEXT), np.float32)
    ctx[1:] = x[:-1, -CONTEXT:]
    return np.concatenate([ctx, x], axis=1)


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _conv1d_k3(x: np.ndarray, w: np.ndarray, b: np.ndarray, stride: int) -> np.ndarray:
    """x [N, Cin, L], w [Cout, Cin, 3], zero pad 1 -> [N, Cout, (L + 2 - 3)//stride + 1]."""
    n, _cin, length = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1)))
    lout = (length + 2 - 3) // stride + 1
    out = np.empty((n, w.shape[0], lout), x.dtype)
    for t in range(lout):
        seg = xp[:, :, t * stride: t * stride + 3]                    # [N, Cin, 3]
        out[:, :, t] = np.einsum("ncj,ocj->no", seg, w) + b
    return out


def encode_windows(w: Dict[str, np.ndarray], frames: np.ndarray, dtype=np.float64) -> np.ndarray:
    """[N, 576] -> per-window feature [N, 128]: everything before the recurrent cell (independent per window)."""
    x = np.asarray(frames, dtype)
    x = np.concatenate([x, x[:, -2: -2 - CONTEXT: -1]], axis=1)       # reflect-pad 64 on the right -> [N, 640]
    n_frames = (x.shape[1] - N_FFT) // HOP + 1                        # 4
    basis = w["stft_basis"].astype(dtype)
    seg = np.stack([x[:, t * HOP: t * HOP + N_FFT] for t in range(n_frames)], axis=1)     # [N, 4, 256]
    spec = seg @ basis.T                                               # [N, 4, 258]
    mag = np.sqrt(spec[..., :N_BINS] ** 2 + spec[..., N_BINS:] ** 2).transpose(0, 2, 1)    # [N, 129, 4]
    y = mag
    for i, (_cin, _cout, stride) in enumerate(ENC_CHANNELS):
        y = np.maximum(_conv1d_k3(y, w[f"enc{i}_w"].astype(dtype), w[f"enc{i}_b"].astype(dtype), stride), 0.0)
    assert y.shape[1:] == (HIDDEN, 1)
    return y[:, :, 0]


def lstm_probs(w: Dict[str, np.ndarray], feats: np.ndarray, state: Optional[Tuple[np.ndarray, np.ndarray]] = None,
               dtype=np.float64):
    """feats [N, 128] in time order -> probs [N], final (h, c). The only sequential part of the network."""
    w_ih, w_hh = w["lstm_w_ih"].astype(dtype), w["lstm_w_hh"].astype(dtype)
    bias = w["lstm_b_ih"].astype(dtype) + w["lstm_b_hh"].astype(dtype)
    out_w, out_b = w["out_w"].astype(dtype), dtype(w["out_b"][0])
    h = np.zeros(HIDDEN, dtype) if state is None else np.asarray(state[0], dtype)
    c = np.zeros(HIDDEN, dtype) if state is None else np.asarray(state[1], dtype)
    gx = np.asarray(feats, dtype) @ w_ih.T + bias
    probs = np.empty(gx.shape[0], dtype)
    for t in range(gx.shape[0]):
        g = gx[t] + w_hh @ h
        i, f, gg, o = _sigmoid(g[:HIDDEN]), _sigmoid(g[HIDDEN: 2 * HIDDEN]), np.tanh(g[2 * HIDDEN: 3 * HIDDEN]), _sigmoid(g[3 * HIDDEN:])
        c = f * c + i * gg
        h = o * np.tanh(c)
        probs[t] = _sigmoid(out_w @ np.maximum(h, 0.0) + out_b)
    return probs.astype(np.float32), (h, c)


def speech_probs(w: Dict[str, np.ndarray], audio: np.ndarray, dtype=np.float64) -> np.ndarray:
    """f32[n] -> f32[ceil(n/512)] speech probability per 32 ms window, fresh state."""
    frames = frame_windows(audio)
    if frames.shape[0] == 0:
        return np.zeros(0, np.float32)
    return lstm_probs(w, encode_windows(w, frames, dtype), None, dtype)[0]


def speech_probs_torch(w: Dict[str, np.ndarray], audio: np.ndarray) -> np.ndarray:
    """The same stack built from torch.nn.functional pieces (conv1d / reflect pad / LSTMCell arithmetic) in float64 —
    an independent build used to check the numpy restatement above."""
    import torch
    import torch.nn.functional as F
    t = lambda a: torch.from_numpy(np.asarray(a)).double()
    frames = frame_windows(audio)
    if frames.shape[0] == 0:
        return np.zeros(0, np.float32)
    x = F.pad(t(frames)[:, None, :], (0, CONTEXT), mode="reflect")
    spec = F.conv1d(x, t(w["stft_basis"])[:, None, :], stride=HOP)              # [N, 258, 4]
    y = torch.sqrt(spec[:, :N_BINS] ** 2 + spec[:, N_BINS:] ** 2)
    for i, (_ci, _co, stride) in enumerate(ENC_CHANNELS):
        y = F.relu(F.conv1d(y, t(w[f"enc{i}_w"]), t(w[f"enc{i}_b"]), stride=stride, padding=1))
    feats = y[:, :, 0]
    cell = torch.nn.LSTMCell(HIDDEN, HIDDEN).double()
    with torch.no_grad():
        cell.weight_ih.copy_(t(w["lstm_w_ih"])); cell.weight_hh.copy_(t(w["lstm_w_hh"]))
        cell.bias_ih.copy_(t(w["lstm_b_ih"])); cell.bias_hh.copy_(t(w["lstm_b_hh"]))
        h = torch.zeros(1, HIDDEN).double(); c = torch.zeros(1, HIDDEN).double()
        probs = []
        for k in range(feats.shape[0]):
            h, c = cell(feats[k: k + 1], (h, c))
            probs.append(torch.sigmoid(F.relu(h) @ t(w["out_w"]) + float(w["out_b"][0])))
    return torch.cat(probs).numpy().astype(np.float32)


from whisperlive_amd.synthetic import energy_following_vad_weights as energy_following_weights  # noqa: E402,F401  (generator, not oracle: kept importable from here for the tests)
