"""Synthetic inputs for benchmarks, tests and the smoke run: seeded PCM and a seeded VAD that really gates.

Generators only — nothing here computes a result the path is judged on. (Until round 6 these two functions lived under
``oracle/``; bench.py imported them from there on its timing path, which blurred the rule that only the ``cpu_baseline`` leg
touches the oracle. They are plain numpy and belong to the package.)"""
from __future__ import annotations

from typing import Dict

import numpy as np

SAMPLE_RATE = 16000


def speech_like_pcm(seconds: float, seed: int = 1234) -> np.ndarray:
    """Deterministic synthetic 'speech-like' PCM of SURVEY.md §8(d): formants x 4 Hz syllabic envelope
    x on/off phrases (2.5 s on / 1.0 s off) + noise, peak 0.5, float32, 16 kHz."""
    rng = np.random.default_rng(seed)
    n = int(round(seconds * SAMPLE_RATE))
    t = np.arange(n) / SAMPLE_RATE
    sig = np.sin(2 * np.pi * 120 * t)
    for f, a in ((700, 0.6), (1200, 0.4), (2600, 0.25)):
        sig = sig + a * np.sin(2 * np.pi * f * t + rng.uniform(0, 2 * np.pi))
    env = 0.5 * (1 + np.sin(2 * np.pi * 4 * t))
    phrase = ((t % 3.5) < 2.5).astype(np.float64)
    sig = sig * env * phrase + rng.normal(0, 0.01, n)
    sig = 0.5 * sig / np.max(np.abs(sig))
    return sig.astype(np.float32)


# Silero-VAD v5 shapes, 16 kHz branch (whisperlive_amd/vad.py, csrc/vad.hip; contract whisper_live/vad.py:50-109)
_VAD_N_FFT = 256
_VAD_HIDDEN = 128
_VAD_ENC_CHANNELS = ((129, 128), (128, 64), (64, 64), (64, 128))     # (Cin, Cout)


def vad_fourier_basis() -> np.ndarray:
    """The fixed STFT filter bank of the Silero front end: Hann-windowed cos / -sin rows, real part first [258, 256]."""
    n = np.arange(_VAD_N_FFT)
    k = np.arange(_VAD_N_FFT // 2 + 1)[:, None]
    win = 0.5 - 0.5 * np.cos(2 * np.pi * n / _VAD_N_FFT)           # periodic Hann
    ang = 2 * np.pi * k * n / _VAD_N_FFT
    return np.concatenate([np.cos(ang) * win, -np.sin(ang) * win], axis=0).astype(np.float32)


def energy_following_vad_weights(seed: int = 0, on_level: float = 0.05) -> Dict[str, np.ndarray]:
    """Seeded weights of the exact Silero shapes whose output FOLLOWS the short-time spectral energy of the input, so that a
    benchmark's VAD gate really gates (speech-like stretches pass, noise-only stretches are cut) while costing exactly what
    the real network costs — no Silero weight file exists offline. Construction: non-negative averaging taps in the four
    convolutions (feature ~ mean STFT magnitude of the window), an LSTM cell opened wide (input / output gates biased on,
    forget gate biased off, small seeded recurrent weights) whose candidate gate is tanh(k (m - on_level)), and a positive
    read-out: p ~ sigmoid(+4) where the mean magnitude m is well above `on_level`, sigmoid(-3) below it."""
    H = _VAD_HIDDEN
    rng = np.random.default_rng(seed)
    w = {"stft_basis": vad_fourier_basis()}
    for i, (cin, cout) in enumerate(_VAD_ENC_CHANNELS):
        w[f"enc{i}_w"] = ((1.0 + 0.3 * rng.uniform(-1, 1, (cout, cin, 3))) / (3.0 * cin)).astype(np.float32)
        w[f"enc{i}_b"] = np.zeros(cout, np.float32)
    # a window's feature after the four averaging layers is ~ 0.3-0.6 x its mean magnitude (zero padding at the frame edges)
    k = 6.0 / on_level
    w_ih = np.zeros((4 * H, H), np.float32)
    w_ih[2 * H: 3 * H] = (k / H) * (1.0 + 0.3 * rng.uniform(-1, 1, (H, H)))
    b = np.zeros(4 * H, np.float32)
    b[:H] = 4.0                      # input gate open
    b[H: 2 * H] = -4.0               # forget gate closed: c = i * g
    b[2 * H: 3 * H] = -0.45 * k * on_level
    b[3 * H:] = 4.0                  # output gate open
    w["lstm_w_ih"] = w_ih
    w["lstm_w_hh"] = (rng.uniform(-1, 1, (4 * H, H)) * 0.02).astype(np.float32)
    w["lstm_b_ih"] = b
    w["lstm_b_hh"] = np.zeros(4 * H, np.float32)
    w["out_w"] = ((7.0 / (0.76 * H)) * (1.0 + 0.2 * rng.uniform(-1, 1, H))).astype(np.float32)
    w["out_b"] = np.asarray([-3.0], np.float32)
    return w
