"""Server metrics: the reference's Prometheus surface (whisper_live/metrics.py — same metric names, labels, helper
functions and `start_metrics_server`, so existing dashboards and the reference's own tests/test_metrics.py apply) plus
an in-process record of the two hot-path observations that DEFINE the benchmark metric (fed at
whisper_live/backend/base.py:130-131): per-chunk transcription latency and seconds of audio processed.
xRT = sum(audio seconds) / sum(latency); p50 chunk latency = median(latency) — `snapshot()`.

prometheus_client is optional, exactly as in the reference: without it the helpers only feed the in-process record."""
from __future__ import annotations

import logging
import threading
from collections import deque

LATENCY_BUCKETS_S = (0.05, 0.1, 0.25, 0.5, 1.0, 2.5, 5.0, 10.0)
# (python name, kind, exported name, label names) — the exported names and labels are the reference's, so a dashboard or
# alert rule written for whisper_live keeps working; python names are what whisper_live.metrics exposes to its tests
_SPEC = (
    ("CONNECTIONS_TOTAL", "counter", "whisperlive_connections_total", ()),
    ("CONNECTIONS_ACTIVE", "gauge", "whisperlive_connections_active", ()),
    ("CONNECTIONS_REJECTED", "counter", "whisperlive_connections_rejected_total", ("reason",)),
    ("TRANSCRIPTION_LATENCY", "histogram", "whisperlive_transcription_latency_seconds", ()),
    ("AUDIO_PROCESSED", "counter", "whisperlive_audio_processed_seconds_total", ()),
    ("SEGMENTS_EMITTED", "counter", "whisperlive_segments_emitted_total", ("completed",)),
    ("REST_REQUESTS", "counter", "whisperlive_rest_requests_total", ("endpoint", "status")),
    ("ERRORS", "counter", "whisperlive_errors_total", ("type",)),
)
try:
    import prometheus_client as _prom
    from prometheus_client import start_http_server

    for _py, _kind, _name, _labels in _SPEC:
        _doc = f"{_name} ({_kind}; MI355X WhisperLive-protocol server)"
        if _kind == "histogram":
            globals()[_py] = _prom.Histogram(_name, _doc, list(_labels), buckets=LATENCY_BUCKETS_S)
        else:
            globals()[_py] = (_prom.Gauge if _kind == "gauge" else _prom.Counter)(_name, _doc, list(_labels))
    _AVAILABLE = True
except ImportError:  # pragma: no cover
    _AVAILABLE = False

# ---- in-process record (always on) -------------------------------------------------------------------------------------
_lock = threading.Lock()
# running sums for xRT (exact over the whole run) + a bounded window of recent latencies for p50 / p95: a server that
# stays up for weeks must not grow by one float per chunk
WINDOW = 65536
_latencies: "deque[float]" = deque(maxlen=WINDOW)
_totals = {"chunks": 0, "latency_s": 0.0, "audio_s": 0.0}
_errors = {}
_segments = {"completed": 0, "partial": 0}
_connections = {"opened": 0, "closed": 0, "active": 0, "rejected": 0}
_server_started = False


def is_available() -> bool:
    return _AVAILABLE


def start_metrics_server(port: int = 9091):
    """Prometheus scrape endpoint on `port` (once per process); logs instead of raising when it cannot start."""
    global _server_started
    if not _AVAILABLE:
        logging.warning("prometheus_client not installed; metrics endpoint disabled")
        return
    try:
        start_http_server(port)
        _server_started = True
        logging.info(f"Prometheus metrics available at http://0.0.0.0:{port}/metrics")
    except Exception as e:  # noqa: BLE001
        logging.error(f"Failed to start metrics server: {e}")


def track_connection_opened():
    with _lock:
        _connections["opened"] += 1
        _connections["active"] += 1
    if _AVAILABLE:
        CONNECTIONS_TOTAL.inc()
        CONNECTIONS_ACTIVE.inc()


def track_connection_closed():
    with _lock:
        _connections["closed"] += 1
        _connections["active"] = max(0, _connections["active"] - 1)
    if _AVAILABLE:
        CONNECTIONS_ACTIVE.dec()


def track_connection_rejected(reason: str = "full"):
    with _lock:
        _connections["rejected"] += 1
    if _AVAILABLE:
        CONNECTIONS_REJECTED.labels(reason=reason).inc()


def track_transcription_latency(seconds: float):
    with _lock:
        _latencies.append(float(seconds))
        _totals["chunks"] += 1
        _totals["latency_s"] += float(seconds)
    if _AVAILABLE:
        TRANSCRIPTION_LATENCY.observe(seconds)


def track_audio_processed(seconds: float):
    with _lock:
        _totals["audio_s"] += float(seconds)
    if _AVAILABLE:
        AUDIO_PROCESSED.inc(seconds)


def track_segment_emitted(completed: bool = True):
    with _lock:
        _segments["completed" if completed else "partial"] += 1
    if _AVAILABLE:
        SEGMENTS_EMITTED.labels(completed=str(bool(completed)).lower()).inc()


def track_rest_request(endpoint: str = "/v1/audio/transcriptions", status="200"):
    if _AVAILABLE:
        REST_REQUESTS.labels(endpoint=endpoint, status=str(status)).inc()


def track_error(error_type: str = "transcription"):
    with _lock:
        _errors[error_type] = _errors.get(error_type, 0) + 1
    if _AVAILABLE:
        ERRORS.labels(type=error_type).inc()


def snapshot(reset: bool = False) -> dict:
    """xRT over everything recorded since the last reset; p50 / p95 over the most recent WINDOW chunks."""
    import statistics
    with _lock:
        lat = list(_latencies)
        tot = dict(_totals)
        out = dict(chunks=tot["chunks"], audio_s=tot["audio_s"], latency_s=tot["latency_s"], errors=dict(_errors),
                   segments=dict(_segments), connections=dict(_connections))
        if reset:
            _latencies.clear(); _errors.clear()
            _totals.update(chunks=0, latency_s=0.0, audio_s=0.0)
            _segments["completed"] = _segments["partial"] = 0
            for k in _connections:
                _connections[k] = 0
    lat.sort()                                                       # outside the lock
    out["xrt"] = (tot["audio_s"] / tot["latency_s"]) if tot["chunks"] and tot["latency_s"] > 0 else None
    out["p50_latency_s"] = statistics.median(lat) if lat else None
    out["p95_latency_s"] = lat[int(0.95 * (len(lat) - 1))] if lat else None
    return out
