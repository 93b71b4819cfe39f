"""CPU oracle for the Whisper log-mel front-end (numpy). Test infrastructure only — see oracle/__init__.py.

Follows faster_whisper.feature_extractor.FeatureExtractor (faster-whisper 1.2.0) as the reference
calls it at whisper_live/transcriber/transcriber_faster_whisper.py:862 (``padding=160``), and the in-tree
restatement whisper_live/transcriber/tensorrt_utils.py:177-190.
"""
from __future__ import annotations

import numpy as np

SAMPLE_RATE = 16000
N_FFT = 400
HOP = 160
N_FRAMES = 3000  # nb_max_frames of a 30 s window


def mel_filters(n_mels: int, dtype=np.float32) -> np.ndarray:
    """Slaney-scale, Slaney-normalised triangular filterbank [n_mels, 201] for sr=16 kHz, n_fft=400.

    Same construction as FeatureExtractor.get_mel_filters (librosa.filters.mel); equals OpenAI's
    mel_filters.npz (tensorrt_utils.py:107-127) and HF audio_utils.mel_filter_bank(..., "slaney", "slaney").
    """
    weights = np.zeros((n_mels, 1 + N_FFT // 2), dtype=dtype)
    fftfreqs = np.fft.rfftfreq(n=N_FFT, d=1.0 / SAMPLE_RATE)
    mels = np.linspace(0.0, 45.245640471924965, n_mels + 2)
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    log_t = mels >= min_log_mel
    freqs[log_t] = min_log_hz * np.exp(logstep * (mels[log_t] - min_log_mel))
    fdiff = np.diff(freqs)
    ramps = np.subtract.outer(freqs, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (freqs[2 : n_mels + 2] - freqs[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def _frames(x: np.ndarray) -> np.ndarray:
    """center=True / reflect-padded framing of torch.stft: [n_frames, 400] windows, hop 160."""
    padded = np.pad(x, (N_FFT // 2, N_FFT // 2), mode="reflect")
    n_frames = 1 + (padded.shape[0] - N_FFT) // HOP
    idx = np.arange(N_FFT)[None, :] + HOP * np.arange(n_frames)[:, None]
    return padded[idx]


def log_mel_spectrogram(waveform: np.ndarray, n_mels: int = 80, padding: int = 160,
                        precise: bool = True) -> np.ndarray:
    """float32 [n_mels, (n + padding) // 160] log-mel features.

    ``precise=True`` evaluates the STFT / mel projection in float64 and rounds once at the end (the
    oracle the HIP kernel is compared against); ``precise=False`` mimics the reference's float32
    pipeline step by step (window f32, rfft on f32, f32 matmul) to bound how far the reference itself
    sits from the exact answer.
    """
    x = np.asarray(waveform, dtype=np.float32)
    if padding:
        x = np.pad(x, (0, padding))
    if x.shape[0] <= N_FFT // 2:
        # np.pad(reflect) wider than the signal reflects repeatedly; emulate it explicitly
        L = x.shape[0]
        idx = np.arange(-(N_FFT // 2), L + N_FFT // 2)
        if L > 1:
            period = 2 * (L - 1)
            idx = np.mod(idx, period)
            idx = np.where(idx >= L, period - idx, idx)
        else:
            idx = np.zeros_like(idx)
        padded = x[idx]
        n_frames = 1 + (padded.shape[0] - N_FFT) // HOP
        fr = padded[np.arange(N_FFT)[None, :] + HOP * np.arange(n_frames)[:, None]]
    else:
        fr = _frames(x)
    window = np.hanning(N_FFT + 1)[:-1]
    if precise:
        spec = np.fft.rfft(fr.astype(np.float64) * window[None, :], axis=-1)
        power = (spec.real ** 2 + spec.imag ** 2)[:-1].T            # drop last frame -> [201, T]
        mel = mel_filters(n_mels).astype(np.float64) @ power
        log_spec = np.log10(np.maximum(mel, 1e-10))
        log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
        return ((log_spec + 4.0) / 4.0).astype(np.float32)
    w32 = window.astype(np.float32)
    spec = np.fft.rfft(fr * w32[None, :], axis=-1).astype(np.complex64)
    power = (np.abs(spec[:-1]) ** 2).T.astype(np.float32)
    mel = mel_filters(n_mels) @ power
    log_spec = np.log10(np.clip(mel, a_min=1e-10, a_max=None))
    log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
    return ((log_spec + 4.0) / 4.0).astype(np.float32)


def pad_or_trim(array: np.ndarray, length: int = N_FRAMES, axis: int = -1) -> np.ndarray:
    """faster_whisper.audio.pad_or_trim (tensorrt_utils.py:80-104): right-pad with ZEROS / trim."""
    if array.shape[axis] > length:
        array = array.take(indices=range(length), axis=axis)
    if array.shape[axis] < length:
        pad_widths = [(0, 0)] * array.ndim
        pad_widths[axis] = (0, length - array.shape[axis])
        array = np.pad(array, pad_widths)
    return array


from whisperlive_amd.synthetic import speech_like_pcm  # noqa: E402,F401  (generator, not oracle: kept importable from here for the tests)
