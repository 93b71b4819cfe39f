"""The two hot-path observations that DEFINE the metric (whisper_live/metrics.py:32-40,100-107, fed at
whisper_live/backend/base.py:130-131): per-chunk transcription latency and seconds of audio processed.
xRT = sum(audio seconds) / sum(latency); p50 chunk latency = median(latency). Kept in-process (thread-safe lists);
mirrored to prometheus_client when it is importable, exactly the optional behaviour of the reference."""
from __future__ import annotations

import threading
from typing import List

_lock = threading.Lock()
_latencies: List[float] = []
_audio_seconds: List[float] = []
_errors = {}
_segments = {"completed": 0, "partial": 0}
_connections = {"opened": 0, "closed": 0, "active": 0, "rejected": 0}

try:  # optional, like the reference
    from prometheus_client import Counter, Histogram
    _H_LAT = Histogram("wlx_transcription_latency_seconds", "Time spent transcribing one audio chunk",
                       buckets=[0.05, 0.1, 0.25, 0.5, 1.0, 2.5, 5.0, 10.0])
    _C_AUDIO = Counter("wlx_audio_seconds_processed_total", "Total seconds of audio transcribed")
    _C_ERR = Counter("wlx_errors_total", "Errors by type", ["type"])
except Exception:  # pragma: no cover
    _H_LAT = _C_AUDIO = _C_ERR = None


def track_transcription_latency(seconds: float):
    with _lock:
        _latencies.append(float(seconds))
    if _H_LAT is not None:
        _H_LAT.observe(seconds)


def track_audio_processed(seconds: float):
    with _lock:
        _audio_seconds.append(float(seconds))
    if _C_AUDIO is not None:
        _C_AUDIO.inc(seconds)


def track_error(kind: str):
    with _lock:
        _errors[kind] = _errors.get(kind, 0) + 1
    if _C_ERR is not None:
        _C_ERR.labels(type=kind).inc()


def track_segment_emitted(completed: bool):
    with _lock:
        _segments["completed" if completed else "partial"] += 1


def track_connection_opened():
    """whisper_live/metrics.py:45-49 (active-connections gauge + total counter)."""
    with _lock:
        _connections["opened"] += 1
        _connections["active"] += 1


def track_connection_closed():
    with _lock:
        _connections["closed"] += 1
        _connections["active"] = max(0, _connections["active"] - 1)


def track_connection_rejected(reason: str = "full"):
    with _lock:
        _connections["rejected"] += 1


def start_metrics_server(port: int = 9091):
    """Prometheus scrape endpoint (whisper_live/metrics.py:29-42); a no-op when prometheus_client is missing."""
    try:
        from prometheus_client import start_http_server
        start_http_server(port)
        return True
    except Exception:  # pragma: no cover
        return False


def snapshot(reset: bool = False) -> dict:
    import statistics
    with _lock:
        lat, aud = list(_latencies), list(_audio_seconds)
        out = dict(chunks=len(lat), audio_s=sum(aud), latency_s=sum(lat), errors=dict(_errors), segments=dict(_segments),
                   connections=dict(_connections),
                   xrt=(sum(aud) / sum(lat)) if lat and sum(lat) > 0 else None,
                   p50_latency_s=statistics.median(lat) if lat else None,
                   p95_latency_s=(sorted(lat)[int(0.95 * (len(lat) - 1))] if lat else None))
        if reset:
            _latencies.clear(); _audio_seconds.clear(); _errors.clear()
            _segments["completed"] = _segments["partial"] = 0
            for k in _connections:
                _connections[k] = 0
    return out
