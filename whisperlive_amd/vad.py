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
  pluggable: ``SileroHIPModel`` runs the network on the MI355X through libwlx.so (``wlx_vad_*``, csrc/vad.hip) from
  a weight archive (``load_silero_npz``; names/shapes in include/wlx.h); ``SileroOnnxModel`` loads an ONNX file
  through onnxruntime when both exist; ``EnergyGateModel`` is a clearly-labelled stand-in (spectral-energy sigmoid)
  that keeps the gate on the path for plumbing and benchmarks. No parity claim is made for the stand-in.
"""
from __future__ import annotations

import bisect
import logging
import os
import threading
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


SILERO_SHAPES = {
    "stft_basis": (258, 256),
    "enc0_w": (128, 129, 3), "enc0_b": (128,), "enc1_w": (64, 128, 3), "enc1_b": (64,),
    "enc2_w": (64, 64, 3), "enc2_b": (64,), "enc3_w": (128, 64, 3), "enc3_b": (128,),
    "lstm_w_ih": (512, 128), "lstm_w_hh": (512, 128), "lstm_b_ih": (512,), "lstm_b_hh": (512,),
    "out_w": (128,), "out_b": (1,),
}


def check_silero_weights(weights: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Validate names / shapes and return C-contiguous float32 copies. Conv weights exported as [Cout, Cin, 3] or, for
    the STFT filter bank and the output layer, with the singleton conv axes torch keeps ([258,1,256], [1,128,1])."""
    out = {}
    for name, shape in SILERO_SHAPES.items():
        if name not in weights:
            raise ValueError(f"Silero VAD weights: '{name}' is missing")
        a = np.asarray(weights[name], dtype=np.float32)
        if a.size != int(np.prod(shape)):
            raise ValueError(f"Silero VAD weights: '{name}' has shape {a.shape}, expected {shape}")
        out[name] = np.ascontiguousarray(a.reshape(shape))
    return out


def load_silero_npz(path: str) -> Dict[str, np.ndarray]:
    with np.load(path) as z:
        return check_silero_weights({k: z[k] for k in z.files})


class SileroHIPModel:
    """Silero VAD on the GPU: padded audio (a multiple of 512 samples) -> one probability per window, through
    ``wlx_vad_probs``. No CPU fallback: construction raises if libwlx.so is missing."""

    def __init__(self, weights: Dict[str, np.ndarray], device: int = 0):
        import ctypes as C
        from . import _lib
        self._lib = _lib
        self.lib = _lib.load()
        self._w = check_silero_weights(weights)              # keeps the host arrays alive during creation
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        cw = _lib.wlx_vad_weights()
        cw.stft_basis = fp(self._w["stft_basis"])
        for i in range(4):
            cw.enc_w[i] = fp(self._w[f"enc{i}_w"])
            cw.enc_b[i] = fp(self._w[f"enc{i}_b"])
        for k in ("lstm_w_ih", "lstm_w_hh", "lstm_b_ih", "lstm_b_hh", "out_w", "out_b"):
            setattr(cw, k, fp(self._w[k]))
        h = C.c_void_p()
        _lib.check(self.lib.wlx_vad_create(C.byref(cw), int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)
        self.last_device_ms = 0.0

    def __call__(self, padded_audio: np.ndarray) -> np.ndarray:
        import ctypes as C
        x = np.ascontiguousarray(padded_audio, dtype=np.float32).reshape(-1)
        cap = x.shape[0] // WINDOW + 1
        probs = np.empty(cap, np.float32)
        n_out, ms = C.c_int32(0), C.c_float(0.0)
        f32p = C.POINTER(C.c_float)
        self._lib.check(self.lib.wlx_vad_probs(self.handle, x.ctypes.data_as(f32p), x.shape[0], probs.ctypes.data_as(f32p),
                                               cap, C.byref(n_out), C.byref(ms)))
        self.last_device_ms = float(ms.value)
        return probs[: n_out.value].copy()

    def probs_resident(self, ring, start: int, n: int) -> np.ndarray:
        """The probabilities `self(np.pad(x, (0, 512 - n % 512)))` would return for x = ring samples [start, start + n) — computed from the
        device-resident ring (whisperlive_amd.engine.PcmRing), no host-to-device copy (include/wlx.h wlx_vad_probs_resident)."""
        import ctypes as C
        n = int(n)
        extra = 1 if n % WINDOW == 0 else 0            # faster_whisper.vad pads to the NEXT multiple: a whole zero window when n is one already
        cap = n // WINDOW + 2
        probs = np.empty(cap, np.float32)
        n_out, ms = C.c_int32(0), C.c_float(0.0)
        f32p = C.POINTER(C.c_float)
        self._lib.check(self.lib.wlx_vad_probs_resident(self.handle, ring._h, int(start), n, extra, probs.ctypes.data_as(f32p), cap,
                                                        C.byref(n_out), C.byref(ms)))
        self.last_device_ms = float(ms.value)
        return probs[: n_out.value].copy()

    def close(self):
        if getattr(self, "handle", None):
            self.lib.wlx_vad_destroy(self.handle)
            self.handle = None

    # no __del__: at interpreter shutdown the HIP runtime may already be gone; close() is explicit, and a process that
    # exits without it simply returns the device memory with its context


_default_model: Optional[Callable[[np.ndarray], np.ndarray]] = None      # installed explicitly (set_default_model / configure)
_default_weights = None                                                   # Silero weights of the process (loaded once)
_device_models: Dict[int, Callable[[np.ndarray], np.ndarray]] = {}       # one SileroHIPModel per GPU (clients shard over GPUs)
_default_lock = threading.Lock()
_resolve_lock = threading.Lock()        # ONE look-up of the process's Silero weights at a time (it may download for up to 10 s)
_resolve_failed: Optional[str] = None   # message of a look-up that found nothing: later calls raise at once instead of trying again


class VadUnavailable(RuntimeError):
    """use_vad was requested but no Silero weights are configured (and the stand-in was not opted into)."""


def set_default_model(model: Optional[Callable[[np.ndarray], np.ndarray]]):
    global _default_model, _default_weights, _resolve_failed
    with _default_lock:
        _default_model = model
        if model is None:
            _default_weights = None
            _resolve_failed = None
            for m in _device_models.values():
                close = getattr(m, "close", None)
                if close:
                    close()
            _device_models.clear()


def configure(weights_path: Optional[str] = None, device: int = 0) -> Callable[[np.ndarray], np.ndarray]:
    """Install the process-wide VAD weights from a file: ``.npz`` (names / shapes of SILERO_SHAPES) or the
    ``silero_vad.onnx`` the reference downloads (whisper_live/vad.py:112-128) — the latter is converted on the fly by
    ``whisperlive_amd.silero_export`` (stdlib protobuf walk, no onnx / onnxruntime needed). The network runs on the GPU the
    calling transcriber lives on: ``get_default_model(device)`` builds one SileroHIPModel per device from these weights."""
    global _default_weights, _resolve_failed
    if weights_path is None:
        return get_default_model(device)
    if weights_path.endswith(".npz"):
        w = load_silero_npz(weights_path)
    else:
        from .silero_export import silero_weights_from_onnx
        w = check_silero_weights(silero_weights_from_onnx(weights_path))
    with _default_lock:
        _default_weights = w
        _resolve_failed = None
    logging.info("VAD: Silero weights from %s", weights_path)
    return get_default_model(device)


def _find_default_weights():
    """Where the reference finds its detector (whisperlive_amd/artifacts.py): WLX_SILERO_VAD_NPZ / WLX_SILERO_VAD_ONNX, the reference's
    own cache file ~/.cache/whisper-live/silero_vad.onnx (whisper_live/vad.py:112-128), the ONNX inside an installed faster-whisper
    wheel, then a first-use download to that cache path where the deployment allows one (not under WLX_NO_DOWNLOAD / HF_HUB_OFFLINE).
    A file that is not the Silero network (unknown layer stack) is skipped with a warning, not trusted — that structural check is
    also what stands between a downloaded file and the model (no checksum is pinned: the reference pins none either). Runs WITHOUT
    the model lock: a stalled download must not block the VAD look-ups of transcribers whose model already exists."""
    from . import artifacts
    cands = artifacts.silero_candidates()
    if not cands and artifacts.downloads_allowed():
        got = artifacts.download_silero()
        if got:
            cands = [("onnx", got)]
    for kind, path in cands:
        try:
            if kind == "npz":
                w = load_silero_npz(path)
            else:
                from .silero_export import silero_weights_from_onnx
                w = check_silero_weights(silero_weights_from_onnx(path))
            logging.info("VAD: Silero (HIP) from %s", path)
            return w
        except Exception as e:  # noqa: BLE001 — an unreadable / different file: try the next place
            logging.warning("VAD: %s is not a usable Silero VAD file (%s: %s)", path, type(e).__name__, e)
    return None


def get_default_model(device: Optional[int] = None) -> Callable[[np.ndarray], np.ndarray]:
    """The VAD model for a transcriber on GPU `device` (None: WLX_VAD_DEVICE or 0): the model installed with set_default_model /
    configure, else one SileroHIPModel per device built from the process's Silero weights. The weights are looked up ONCE per
    process (`_find_default_weights`: environment variables, the reference's cache file, an installed wheel, then a download where
    allowed), outside the model lock; a look-up that found nothing is remembered, so every later call raises VadUnavailable
    immediately instead of waiting for another download time-out (configure() / set_default_model(None) reset that). Without
    weights ``use_vad`` FAILS unless WLX_ALLOW_VAD_STANDIN=1 opts into the labelled energy gate — a default deployment must not
    silently gate audio with something that is not the reference's detector."""
    global _default_model, _default_weights, _resolve_failed
    dev = int(os.environ.get("WLX_VAD_DEVICE", "0")) if device is None else int(device)
    with _default_lock:
        if _default_model is not None:
            return _default_model
        if dev in _device_models:
            return _device_models[dev]
        have = _default_weights is not None
    if not have:
        with _resolve_lock:
            with _default_lock:
                have, failed = _default_weights is not None, _resolve_failed
            if not have and failed is None:
                w = _find_default_weights()
                with _default_lock:
                    if w is not None:
                        _default_weights = w
                    else:
                        _resolve_failed = failed = (
                            "use_vad needs Silero VAD weights: none found in WLX_SILERO_VAD_ONNX / WLX_SILERO_VAD_NPZ, "
                            "~/.cache/whisper-live/silero_vad.onnx (the file the reference downloads) or an installed faster_whisper / "
                            "silero_vad package, and no download was possible; pass --vad_weights to the server, or opt into the "
                            "energy-gate stand-in with WLX_ALLOW_VAD_STANDIN=1")
                    have = w is not None
        if not have:
            if os.environ.get("WLX_ALLOW_VAD_STANDIN") == "1":
                with _default_lock:
                    if _default_model is None:
                        logging.warning("VAD: WLX_ALLOW_VAD_STANDIN=1 — using the energy-gate stand-in, which is NOT the reference's "
                                        "speech detector (set WLX_SILERO_VAD_ONNX / WLX_SILERO_VAD_NPZ or pass --vad_weights)")
                        _default_model = EnergyGateModel()
                    return _default_model
            raise VadUnavailable(failed)
    with _default_lock:
        if _default_model is not None:
            return _default_model
        if dev not in _device_models:
            _device_models[dev] = SileroHIPModel(_default_weights, dev)
        return _device_models[dev]


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

    # The loop below runs on the host between the VAD launch and the log-mel launch — the GPU waits for it. Iterating a float32 array
    # yields numpy scalars (82 us for the 250 windows of an 8 s chunk, 218 us for 30 s); the same values as Python floats, compared
    # against the thresholds ROUNDED TO float32 — which is what numpy's float32-scalar >= python-float comparison does (NEP 50) —
    # give the same decisions in a third of the time.
    if isinstance(probs, np.ndarray) and probs.dtype == np.float32:
        thr, neg = float(np.float32(thr)), float(np.float32(neg))
        probs = probs.tolist()

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


def speech_segments_from_probs_native(probs: np.ndarray, n_samples: int, opt: VadOptions, sampling_rate: int = 16000) -> List[Dict[str, int]]:
    """`speech_segments_from_probs` through libwlx.so (include/wlx.h wlx_vad_segments: the same loop in C, ~2 us instead of 36-136 us of host
    time that the GPU spends waiting between the gate and the log-mel). Used on the GPU gate's path (the library is loaded there anyway);
    tests/test_vad_segments_native.py holds it to the Python statement case by case."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    p = np.ascontiguousarray(probs, dtype=np.float32)
    thr = opt.threshold
    neg = opt.neg_threshold if opt.neg_threshold is not None else max(thr - 0.15, 0.01)
    pad = sampling_rate * opt.speech_pad_ms / 1000
    cap = p.shape[0] + 2
    out = np.empty((cap, 2), np.int64)
    n_out = C.c_int32(0)
    _lib.check(lib.wlx_vad_segments(p.ctypes.data_as(C.POINTER(C.c_float)), int(p.shape[0]), int(n_samples),
                                    float(np.float32(thr)), float(np.float32(neg)), float(sampling_rate * opt.min_speech_duration_ms / 1000),
                                    float(pad), float(sampling_rate * opt.max_speech_duration_s - WINDOW - 2 * pad),
                                    float(sampling_rate * opt.min_silence_duration_ms / 1000), float(sampling_rate * 98 / 1000),
                                    out.ctypes.data_as(C.POINTER(C.c_int64)), cap, C.byref(n_out)))
    return [{"start": int(a), "end": int(b)} for a, b in out[: n_out.value].tolist()]


def get_speech_timestamps(audio: np.ndarray, vad_options: Optional[VadOptions] = None, sampling_rate: int = 16000,
                          model: Optional[Callable[[np.ndarray], np.ndarray]] = None) -> List[Dict[str, int]]:
    opt = vad_options or VadOptions()
    n = int(audio.shape[0])
    padded = np.pad(audio.astype(np.float32, copy=False), (0, WINDOW - n % WINDOW))
    model = model or get_default_model()
    probs = model(padded)
    if isinstance(model, SileroHIPModel) and isinstance(probs, np.ndarray) and probs.dtype == np.float32:
        return speech_segments_from_probs_native(probs, n, opt, sampling_rate)
    return speech_segments_from_probs(probs, n, opt, sampling_rate)


def get_speech_timestamps_resident(ring, start: int, n: int, vad_options: Optional[VadOptions] = None, sampling_rate: int = 16000,
                                   model=None) -> List[Dict[str, int]]:
    """get_speech_timestamps for audio that lives in a device PCM ring: samples [start, start + n). `model` must be a SileroHIPModel
    (anything else has no device path: the caller falls back to the host audio)."""
    opt = vad_options or VadOptions()
    probs = model.probs_resident(ring, start, n)
    return speech_segments_from_probs_native(probs, int(n), opt, sampling_rate)


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
