"""Voice-activity gate of the transcription path — the role of faster_whisper.vad (faster-whisper 1.2.0, un-vendored)
in the reference: ``get_speech_timestamps`` + ``collect_chunks`` at
whisper_live/transcriber/transcriber_faster_whisper.py:830-838 (options ``{"threshold": 0.5}`` from
whisper_live/backend/faster_whisper_backend.py:85), ``SpeechTimestampsMap`` at :1792-1817.

Two parts:
* the SEGMENTATION logic (hysteresis over per-32 ms speech probabilities, padding, concatenation, time map) — pure
  integer/float bookkeeping, restated here from the published Silero-VAD `get_speech_timestamps` procedure the
  reference's dependency follows (⚠ verify against the wheel the first time one is available);
* the PROBABILITY model (Silero VAD v5: 512-sample windows with 64 samples of left context, LSTM state [2,1,128];
  I/O contract documented in-tree at whisper_live/vad.py:50-109). Its weights are not available offline, so the model is
  pluggable: ``SileroOnnxModel`` loads an ONNX file through onnxruntime when both exist; ``EnergyGateModel`` is a
  clearly-labelled stand-in (spectral-energy sigmoid) that keeps the gate on the path for plumbing and benchmarks.
  No parity claim is made for the stand-in.
"""
from __future__ import annotations

import bisect
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

WINDOW = 512          # samples per probability (32 ms at 16 kHz)
CONTEXT = 64


@dataclass
class VadOptions:
    threshold: float = 0.5
    neg_threshold: Optional[float] = None
    min_speech_duration_ms: int = 0
    max_speech_duration_s: float = float("inf")
    min_silence_duration_ms: int = 2000
    speech_pad_ms: int = 400


class EnergyGateModel:
    """STAND-IN probability model (NOT Silero): p = sigmoid(k * (10*log10(window power) - level_db)).
    Deterministic, stateless; used only when no Silero weights can be loaded."""

    def __init__(self, level_db: float = -45.0, slope: float = 0.6):
        self.level_db, self.slope = level_db, slope

    def __call__(self, padded_audio: np.ndarray) -> np.ndarray:
        w = padded_audio.reshape(-1, WINDOW).astype(np.float64)
        power = (w * w).mean(axis=1) + 1e-12
        return (1.0 / (1.0 + np.exp(-self.slope * (10.0 * np.log10(power) - self.level_db)))).astype(np.float32)


class SileroOnnxModel:
    """Silero VAD v5 through onnxruntime (inputs `input` [1, 576], `state` [2,1,128], `sr` int64 -> `output`, `stateN`),
    the same contract as whisper_live/vad.py:56-109. Raises at construction if onnxruntime or the file is missing."""

    def __init__(self, path: str):
        import onnxruntime as ort  # noqa: F401  (absent offline -> ImportError surfaces to the caller)
        opts = ort.SessionOptions()
        opts.inter_op_num_threads = 1
        opts.intra_op_num_threads = 1
        self.session = ort.InferenceSession(path, providers=["CPUExecutionProvider"], sess_options=opts)

    def __call__(self, padded_audio: np.ndarray) -> np.ndarray:
        state = np.zeros((2, 1, 128), dtype=np.float32)
        ctx = np.zeros(CONTEXT, dtype=np.float32)
        sr = np.array(16000, dtype=np.int64)
        probs = []
        for w in padded_audio.reshape(-1, WINDOW).astype(np.float32):
            x = np.concatenate([ctx, w])[None, :]
            out, state = self.session.run(None, {"input": x, "state": state, "sr": sr})
            probs.append(float(np.asarray(out).reshape(-1)[0]))
            ctx = w[-CONTEXT:]
        return np.asarray(probs, dtype=np.float32)


_default_model: Optional[Callable[[np.ndarray], np.ndarray]] = None


def set_default_model(model: Optional[Callable[[np.ndarray], np.ndarray]]):
    global _default_model
    _default_model = model


def get_default_model() -> Callable[[np.ndarray], np.ndarray]:
    global _default_model
    if _default_model is None:
        import os
        path = os.environ.get("WLX_SILERO_VAD_ONNX")
        if path and os.path.isfile(path):
            _default_model = SileroOnnxModel(path)
        else:
            _default_model = EnergyGateModel()
    return _default_model


def speech_segments_from_probs(probs: Sequence[float], n_samples: int, opt: VadOptions, sampling_rate: int = 16000
                               ) -> List[Dict[str, int]]:
    """Hysteresis segmentation of per-window speech probabilities into padded sample ranges."""
    thr = opt.threshold
    neg = opt.neg_threshold if opt.neg_threshold is not None else max(thr - 0.15, 0.01)
    min_speech = sampling_rate * opt.min_speech_duration_ms / 1000
    pad = sampling_rate * opt.speech_pad_ms / 1000
    max_speech = sampling_rate * opt.max_speech_duration_s - WINDOW - 2 * pad
    min_silence = sampling_rate * opt.min_silence_duration_ms / 1000
    min_silence_at_max = sampling_rate * 98 / 1000

    speeches: List[Dict[str, int]] = []
    cur: Dict[str, int] = {}
    active = False
    silence_from = 0          # start of the silence currently being timed (0 = none)
    cut_at = 0                # last long-enough silence start (split point for over-long speech)
    resume_at = 0             # where speech resumed after cut_at
    for i, p in enumerate(probs):
        pos = WINDOW * i
        if p >= thr and silence_from:
            silence_from = 0
            if resume_at < cut_at:
                resume_at = pos
        if p >= thr and not active:
            active = True
            cur["start"] = pos
            continue
        if active and pos - cur["start"] > max_speech:
            if cut_at:
                cur["end"] = cut_at
                speeches.append(cur)
                cur = {}
                if resume_at < cut_at:
                    active = False
                else:
                    cur["start"] = resume_at
                cut_at = resume_at = silence_from = 0
            else:
                cur["end"] = pos
                speeches.append(cur)
                cur = {}
                cut_at = resume_at = silence_from = 0
                active = False
                continue
        if p < neg and active:
            if not silence_from:
                silence_from = pos
            if pos - silence_from > min_silence_at_max:
                cut_at = silence_from
            if pos - silence_from < min_silence:
                continue
            cur["end"] = silence_from
            if cur["end"] - cur["start"] > min_speech:
                speeches.append(cur)
            cur = {}
            cut_at = resume_at = silence_from = 0
            active = False
    if cur and n_samples - cur["start"] > min_speech:
        cur["end"] = n_samples
        speeches.append(cur)

    for i, sp in enumerate(speeches):
        if i == 0:
            sp["start"] = int(max(0, sp["start"] - pad))
        if i != len(speeches) - 1:
            gap = speeches[i + 1]["start"] - sp["end"]
            if gap < 2 * pad:
                sp["end"] += int(gap // 2)
                speeches[i + 1]["start"] = int(max(0, speeches[i + 1]["start"] - gap // 2))
            else:
                sp["end"] = int(min(n_samples, sp["end"] + pad))
                speeches[i + 1]["start"] = int(max(0, speeches[i + 1]["start"] - pad))
        else:
            sp["end"] = int(min(n_samples, sp["end"] + pad))
    return speeches


def get_speech_timestamps(audio: np.ndarray, vad_options: Optional[VadOptions] = None, sampling_rate: int = 16000,
                          model: Optional[Callable[[np.ndarray], np.ndarray]] = None) -> List[Dict[str, int]]:
    opt = vad_options or VadOptions()
    n = int(audio.shape[0])
    padded = np.pad(audio.astype(np.float32, copy=False), (0, WINDOW - n % WINDOW))
    probs = (model or get_default_model())(padded)
    return speech_segments_from_probs(probs, n, opt, sampling_rate)


def collect_chunks(audio: np.ndarray, chunks: List[Dict[str, int]], sampling_rate: int = 16000,
                   max_duration: float = float("inf")) -> Tuple[List[np.ndarray], List[dict]]:
    """Concatenate the speech ranges (one output chunk unless max_duration forces a split). Empty input -> one
    empty chunk, which is what makes transcribe() return (None, None) (transcriber_faster_whisper.py:860-861)."""
    if not chunks:
        return [np.array([], dtype=np.float32)], [{"start_time": 0, "end_time": 0, "segments": []}]
    out_audio: List[np.ndarray] = []
    out_meta: List[dict] = []
    parts: List[np.ndarray] = []
    segs: List[Dict[str, int]] = []
    cur = total = 0
    for ch in chunks:
        ln = ch["end"] - ch["start"]
        if parts and cur + ln > max_duration * sampling_rate:
            out_audio.append(np.concatenate(parts))
            out_meta.append({"start_time": total / sampling_rate, "end_time": (total + cur) / sampling_rate, "segments": segs})
            total += cur
            parts, segs, cur = [], [], 0
        parts.append(audio[ch["start"]:ch["end"]])
        segs.append(ch)
        cur += ln
    out_audio.append(np.concatenate(parts))
    out_meta.append({"start_time": total / sampling_rate, "end_time": (total + cur) / sampling_rate, "segments": segs})
    return out_audio, out_meta


class SpeechTimestampsMap:
    """Maps a time in the VAD-compressed audio back to the original timeline."""

    def __init__(self, chunks: List[Dict[str, int]], sampling_rate: int, time_precision: int = 2):
        self.sampling_rate = sampling_rate
        self.time_precision = time_precision
        self.chunk_end_sample: List[int] = []
        self.total_silence_before: List[float] = []
        prev_end = removed = 0
        for ch in chunks:
            removed += ch["start"] - prev_end
            prev_end = ch["end"]
            self.chunk_end_sample.append(ch["end"] - removed)
            self.total_silence_before.append(removed / sampling_rate)

    def get_chunk_index(self, time: float, is_end: bool = False) -> int:
        sample = int(time * self.sampling_rate)
        if is_end and sample in self.chunk_end_sample:
            return self.chunk_end_sample.index(sample)
        return min(bisect.bisect(self.chunk_end_sample, sample), len(self.chunk_end_sample) - 1)

    def get_original_time(self, time: float, chunk_index: Optional[int] = None, is_end: bool = False) -> float:
        if chunk_index is None:
            chunk_index = self.get_chunk_index(time, is_end)
        return round(self.total_silence_before[chunk_index] + time, self.time_precision)
