"""The WebSocket server shell in front of the hot path (§8a row 1): a protocol-exact
re-creation of ``TranscriptionServer`` / ``ClientManager`` / ``BackendType`` (whisper_live/server.py:45-181,184-196,
288-488,600-690,874-935) so that stock WhisperLive clients connect unchanged and every chunk they stream lands in
``ServeClientHIP`` -> ``libwlx.so``. It is the caller that turns socket bytes into the PCM the engine ingests
(``get_audio_from_websocket``, server.py:365-385) and owns the per-connection lifecycle; there is no arithmetic here
beyond that int -> f32 conversion.

What is the same: the JSON option keys read from the first message (server.py:396-404, 324-346), the WAIT / WARNING /
ERROR / SERVER_READY messages, the ``END_OF_AUDIO`` terminator, the ``audio_format`` rules (float32 default; int16
``/32768`` when ``raw_pcm_input`` or ``audio_format == "int16"``; uint8 ``(x-128)/128``), the connection-time limit,
``max_clients`` with the wait estimate, bearer / ``?token=`` auth (server.py:34-42), ``single_model`` + batch-worker
start-up (server.py:334-344, 660-675), the argument names of ``run``.

What differs, on purpose:
  * the model sits on an MI355X: ``faster_whisper`` (the string stock clients test for, client.py:182,399) and
    ``hip`` both select ``ServeClientHIP``; ``tensorrt`` / ``openvino`` are accepted names that fall back to it with
    the reference's own WARNING message (server.py:253-258, 282-286), since neither stack exists on this hardware;
  * GPUs: ``devices=[...]`` shards connections over GPUs, connection i -> ``devices[i mod n]`` (sharding.assign_gpu;
    SURVEY.md §8e) — one engine + slot pool per GPU, no cross-GPU traffic;
  * translation, diarization and the OpenAI-style REST endpoint are separate products around the path (their own
    models / HTTP stack) and are not provided: the options are accepted and ignored with a log line, ``enable_rest``
    raises;
  * ``use_vad`` is kept per connection (the reference stores it on the server object, server.py:394, so two clients
    with different settings race).
"""
from __future__ import annotations

import functools
import json
import logging
import os
import threading
import time
from enum import Enum
from http import HTTPStatus
from typing import Callable, List, Optional, Sequence
from urllib.parse import parse_qs, urlparse

import numpy as np

from . import metrics as wl_metrics
from . import vad as _vad
from . import ws
from .serve_client import ServeClientBase, ServeClientHIP
from .sharding import assign_gpu
from .ws import ConnectionClosed

MAX_SLOT_BATCH = 64         # clips one engine slot encodes and decodes together (include/wlx.h wlx_slot_create: 64 items x 5 beams = 320 rows)

END_OF_AUDIO = b"END_OF_AUDIO"      # whisper_live/server.py:376, client.py:23
AUDIO_FORMATS = ("float32", "int16", "uint8")


def _websocket_auth(api_key, connection, request):
    """``process_request`` hook: 401 unless ``Authorization: Bearer <key>`` or ``?token=<key>`` (server.py:34-42)."""
    auth = request.headers.get("Authorization", "")
    token_param = None
    if "?" in request.path:
        token_param = parse_qs(urlparse(request.path).query).get("token", [None])[0]
    if auth == f"Bearer {api_key}" or token_param == api_key:
        return None
    return connection.respond(HTTPStatus.UNAUTHORIZED, "Unauthorized\n")


class ClientManager:
    """Connection table with a capacity and a per-connection time limit (server.py:45-158)."""

    def __init__(self, max_clients=4, max_connection_time=600):
        self.clients = {}
        self.start_times = {}
        self.max_clients = max_clients
        self.max_connection_time = max_connection_time
        self.lock = threading.Lock()

    def add_client(self, websocket, client):
        with self.lock:
            self.clients[websocket] = client
            self.start_times[websocket] = time.time()

    def get_client(self, websocket):
        with self.lock:
            return self.clients.get(websocket, False)

    def remove_client(self, websocket):
        with self.lock:
            client = self.clients.pop(websocket, None)
            self.start_times.pop(websocket, None)
        if client:
            client.cleanup()

    def _min_remaining(self) -> Optional[float]:
        now = time.time()
        left = [self.max_connection_time - (now - t0) for t0 in self.start_times.values()]
        return min(left) if left else None

    def get_wait_time(self):
        """Minutes until the connection closest to its limit frees a slot; 0 with no connections."""
        with self.lock:
            left = self._min_remaining()
        return left / 60 if left is not None else 0

    def is_server_full(self, websocket, options):
        with self.lock:
            if len(self.clients) < self.max_clients:
                return False
            left = self._min_remaining()
            websocket.send(json.dumps({"uid": options["uid"], "status": "WAIT",
                                       "message": left / 60 if left is not None else 0}))
            return True

    def is_client_timeout(self, websocket):
        with self.lock:
            elapsed = time.time() - self.start_times[websocket]
            client = self.clients.get(websocket)
        if elapsed >= self.max_connection_time and client:
            client.disconnect()
            logging.warning(f"Client with uid '{client.client_uid}' disconnected due to overtime.")
            return True
        return False


class BackendType(Enum):
    FASTER_WHISPER = "faster_whisper"
    TENSORRT = "tensorrt"
    OPENVINO = "openvino"
    HIP = "hip"

    @staticmethod
    def valid_types() -> List[str]:
        return [b.value for b in BackendType]

    @staticmethod
    def is_valid(backend: str) -> bool:
        return backend in BackendType.valid_types()

    def is_faster_whisper(self) -> bool:
        return self == BackendType.FASTER_WHISPER

    def is_tensorrt(self) -> bool:
        return self == BackendType.TENSORRT

    def is_openvino(self) -> bool:
        return self == BackendType.OPENVINO

    def is_hip(self) -> bool:
        return self == BackendType.HIP


class TranscriptionServer:
    RATE = 16000

    def __init__(self):
        self.client_manager = None
        self.no_voice_activity_chunks = 0
        self.use_vad = True
        self.single_model = False
        self.batch_config = None
        self.raw_pcm_input = False
        self.audio_formats = {}
        self.segment_post_processor = None
        self.backend = BackendType.HIP
        self.cache_path = "~/.cache/whisper-live/"
        self.devices: List[int] = [0]
        self.model_factory: Optional[Callable] = None      # (model, device_index) -> transcriber; tests / embedding
        self._n_connections = 0
        self._conn_lock = threading.Lock()
        self._server: Optional[ws.Server] = None

    # ---- per-connection set-up -----------------------------------------------------------------------------------
    def _next_device(self) -> int:
        with self._conn_lock:
            i = self._n_connections
            self._n_connections += 1
        return self.devices[assign_gpu(i, len(self.devices))]

    def initialize_client(self, websocket, options, faster_whisper_custom_model_path, whisper_tensorrt_path,
                          trt_multilingual, trt_py_session=False):
        if options.get("enable_translation", False):
            logging.warning("enable_translation: the translation side-channel is not part of this server; ignored")
        if options.get("enable_diarization", False):
            logging.warning("enable_diarization: speaker diarization is not part of this server; disabled")

        if self.backend.is_tensorrt() or self.backend.is_openvino():
            name = "TensorRT-LLM" if self.backend.is_tensorrt() else "OpenVINO"
            logging.error(f"{name} not supported on an MI355X server")
            websocket.send(json.dumps({
                "uid": options["uid"], "status": "WARNING",
                "message": f"{name} not supported on Server yet. Reverting to available backend: 'faster_whisper'"}))
            self.backend = BackendType.FASTER_WHISPER

        client: Optional[ServeClientBase] = None
        try:
            if faster_whisper_custom_model_path is not None:
                logging.info(f"Using custom model {faster_whisper_custom_model_path}")
                options["model"] = faster_whisper_custom_model_path
            device_index = self._next_device()
            use_vad = bool(options.get("use_vad"))
            if use_vad and self.model_factory is None:
                # the gate must be the reference's detector or nothing: without Silero weights the client is told and
                # the session runs ungated (WLX_ALLOW_VAD_STANDIN=1 opts into the labelled energy gate instead)
                try:
                    _vad.get_default_model(device_index)
                except _vad.VadUnavailable as e:
                    logging.warning(f"use_vad requested by {options['uid']} but unavailable: {e}")
                    websocket.send(json.dumps({"uid": options["uid"], "status": "WARNING",
                                               "message": "use_vad ignored: no Silero VAD weights are configured on this server"}))
                    use_vad = False
            client = ServeClientHIP(
                websocket,
                language=options["language"],
                task=options["task"],
                client_uid=options["uid"],
                model=options["model"],
                initial_prompt=options.get("initial_prompt"),
                vad_parameters=options.get("vad_parameters"),
                use_vad=use_vad,
                single_model=self.single_model,
                send_last_n_segments=options.get("send_last_n_segments", 10),
                no_speech_thresh=options.get("no_speech_thresh", 0.45),
                clip_audio=options.get("clip_audio", False),
                same_output_threshold=options.get("same_output_threshold", 10),
                cache_path=self.cache_path,
                hotwords=options.get("hotwords"),
                word_timestamps=options.get("word_timestamps", False),
                device_index=device_index,
                model_factory=self.model_factory,
                max_batch=self.batch_config["max_batch_size"] if self.batch_config is not None else 1,
            )
            if not hasattr(client, "transcriber"):          # model load failed: ERROR already sent, socket closed
                return
            logging.info(f"Running HIP backend on GPU {device_index}.")
            # one batch worker per GPU, started once that GPU's shared transcriber exists (server.py:334-344)
            if self.batch_config is not None and self.single_model and device_index not in ServeClientHIP.BATCH_WORKERS:
                from .batching import BatchInferenceWorker
                with ServeClientHIP.MODELS_LOCK:
                    if device_index not in ServeClientHIP.BATCH_WORKERS:
                        worker = BatchInferenceWorker(transcriber=client.transcriber, lanes=getattr(self, "batch_lanes", 2), **self.batch_config)
                        worker.start()
                        ServeClientHIP.BATCH_WORKERS[device_index] = worker
        except Exception as e:  # noqa: BLE001 — same as server.py:345-347: log and leave the connection unregistered
            logging.error(e)
            return

        if self.segment_post_processor is not None:
            client.segment_post_processor = self.segment_post_processor
        self.client_manager.add_client(websocket, client)

    # ---- socket bytes -> PCM ---------------------------------------------------------------
    def get_audio_from_websocket(self, websocket):
        """One packet -> float32 PCM in [-1, 1); ``False`` on the END_OF_AUDIO terminator (server.py:365-385)."""
        frame_data = websocket.recv()
        if frame_data == END_OF_AUDIO:
            return False
        if isinstance(frame_data, str):
            raise ValueError("audio packets must be binary frames")
        audio_format = self.audio_formats.get(websocket)
        if audio_format == "uint8":
            return (np.frombuffer(frame_data, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        if self.raw_pcm_input or audio_format == "int16":
            return np.frombuffer(frame_data, dtype=np.int16).astype(np.float32) / 32768.0
        return np.frombuffer(frame_data, dtype=np.float32)

    def handle_new_connection(self, websocket, faster_whisper_custom_model_path, whisper_tensorrt_path,
                              trt_multilingual, trt_py_session=False):
        try:
            logging.info("New client connected")
            options = json.loads(websocket.recv())
            self.use_vad = options.get("use_vad")
            if self.client_manager.is_server_full(websocket, options):
                wl_metrics.track_connection_rejected(reason="full")
                websocket.close()
                return False
            audio_format = options.get("audio_format", "float32")
            if audio_format not in AUDIO_FORMATS:
                raise ValueError(f"Unsupported audio_format: {audio_format}")
            self.audio_formats[websocket] = audio_format
            self.initialize_client(websocket, options, faster_whisper_custom_model_path, whisper_tensorrt_path,
                                   trt_multilingual, trt_py_session=trt_py_session)
            wl_metrics.track_connection_opened()
            return True
        except json.JSONDecodeError:
            logging.error("Failed to decode JSON from client")
            return False
        except ConnectionClosed:
            logging.info("Connection closed by client")
            return False
        except Exception as e:  # noqa: BLE001
            logging.error(f"Error during new connection initialization: {str(e)}")
            return False

    def process_audio_frames(self, websocket):
        frame_np = self.get_audio_from_websocket(websocket)
        if frame_np is False:
            return False
        client = self.client_manager.get_client(websocket)
        if not client:                                   # initialise failed (model load): nothing to feed
            return False
        client.add_frames(frame_np)
        return True

    def recv_audio(self, websocket, backend: BackendType = BackendType.HIP, faster_whisper_custom_model_path=None,
                   whisper_tensorrt_path=None, trt_multilingual=False, trt_py_session=False):
        """Connection handler: options message, then packets until END_OF_AUDIO / close / time limit
        (server.py:441-488)."""
        self.backend = backend
        if not self.handle_new_connection(websocket, faster_whisper_custom_model_path, whisper_tensorrt_path,
                                          trt_multilingual, trt_py_session=trt_py_session):
            return
        try:
            while self.client_manager.get_client(websocket) and not self.client_manager.is_client_timeout(websocket):
                if not self.process_audio_frames(websocket):
                    break
        except ConnectionClosed:
            logging.info("Connection closed by client")
        except Exception as e:  # noqa: BLE001
            logging.error(f"Unexpected error: {str(e)}")
        finally:
            if self.client_manager.get_client(websocket):
                self.cleanup(websocket)
                websocket.close()
            self.audio_formats.pop(websocket, None)
            wl_metrics.track_connection_closed()

    def cleanup(self, websocket):
        if self.client_manager.get_client(websocket):
            self.client_manager.remove_client(websocket)
        self.audio_formats.pop(websocket, None)

    # ---- entry point ---------------------------------------------------------------------------------------------
    def configure(self, backend="hip", faster_whisper_custom_model_path=None, whisper_tensorrt_path=None,
                  single_model=False, max_clients=4, max_connection_time=600, cache_path="~/.cache/whisper-live/",
                  enable_rest=False, batch_enabled=False, batch_max_size=8, batch_window_ms=50, raw_pcm_input=False,
                  segment_post_processor=None, devices: Optional[Sequence[int]] = None, model_factory=None,
                  vad_weights: Optional[str] = None, batch_lanes: int = 2):
        """Argument validation and server state of ``run`` (server.py:644-690), split out so it can be used without
        opening a socket. ``batch_lanes``: worker lanes of each GPU's BatchInferenceWorker (1 = the reference's single worker
        thread; batching.py)."""
        self.cache_path = cache_path
        self.raw_pcm_input = raw_pcm_input
        if max_clients < 1:
            raise ValueError(f"max_clients must be >= 1, got {max_clients}")
        if max_connection_time <= 0:
            raise ValueError(f"max_connection_time must be > 0, got {max_connection_time}")
        if batch_enabled and batch_max_size < 1:
            raise ValueError(f"batch_max_size must be >= 1, got {batch_max_size}")
        if batch_enabled and batch_window_ms < 0:
            raise ValueError(f"batch_window_ms must be >= 0, got {batch_window_ms}")
        if enable_rest:
            raise NotImplementedError("the OpenAI-style REST endpoint is not part of this server (WebSocket path only)")
        self.segment_post_processor = segment_post_processor
        self.client_manager = ClientManager(max_clients, max_connection_time)
        if faster_whisper_custom_model_path is not None and not os.path.exists(faster_whisper_custom_model_path):
            raise ValueError(f"Custom model '{faster_whisper_custom_model_path}' is not a valid path "
                             "(there is no hub download on this server).")
        if whisper_tensorrt_path is not None and not os.path.exists(whisper_tensorrt_path):
            raise ValueError(f"TensorRT model '{whisper_tensorrt_path}' is not a valid path.")
        if batch_enabled:
            single_model = True                       # batching needs the shared per-GPU transcriber
            if batch_lanes < 1:
                raise ValueError(f"batch_lanes must be >= 1, got {batch_lanes}")
            if batch_max_size > MAX_SLOT_BATCH:
                # the reference takes any max_batch_size (batch_inference.py:113-121); here one engine slot holds 64 clips x 5 beams
                # (round 5: it was 12 clips). Beyond that further clips overlap through --batch_lanes.
                logging.warning(f"--batch_max_size {batch_max_size}: one batch holds at most {MAX_SLOT_BATCH} clips on this backend; using "
                                f"{MAX_SLOT_BATCH} (more lanes, --batch_lanes, are how further clips overlap)")
                batch_max_size = MAX_SLOT_BATCH
            self.batch_config = {"max_batch_size": batch_max_size, "batch_window_ms": batch_window_ms}   # (the reference's two keys)
            self.batch_lanes = int(batch_lanes)
            logging.info(f"Batch inference enabled (max_batch={batch_max_size}, window={batch_window_ms}ms, lanes={batch_lanes})")
        else:
            self.batch_config = None
        # One engine (one copy of the weights) per GPU serves every client on that GPU through its own slot, so the
        # shared model is the natural mode here whatever the flag says for stock models (server.py:664-675).
        self.single_model = bool(single_model)
        if not BackendType.is_valid(backend):
            raise ValueError(f"{backend} is not a valid backend type. Choose backend from {BackendType.valid_types()}")
        self.devices = list(devices) if devices else [0]
        if any(d < 0 for d in self.devices):
            raise ValueError("device indices must be >= 0")
        self.model_factory = model_factory
        # More than four concurrent sessions per GPU on independent slots lose throughput: a device gives at most four slots a
        # hardware queue of their own (libwlx: WLX_DEDICATED_QUEUES), the fifth live slot sends every slot back to the shared queue
        # pool (measured on one MI355X, Whisper-small shapes: 4 streams 2753x real time, 8 streams 2190x), while rows
        # batched into one decode are nearly free (--batch_inference: 12 windows per decode 6050x). Say so instead of degrading
        # silently (task 7); the mode is the operator's choice because the batched path keeps the reference's
        # batch-mode result quirks (first 30 s per request, greedy fallback temperatures: whisper_live/batch_inference.py:259,343).
        per_gpu = -(-int(max_clients) // max(1, len(self.devices)))
        if not batch_enabled and per_gpu > 4:
            logging.warning(f"max_clients={max_clients} over {len(self.devices)} GPU(s) = up to {per_gpu} concurrent sessions per GPU without "
                            "--batch_inference: beyond 4 sessions per GPU the independent decode streams share hardware queues and aggregate "
                            "throughput DROPS (8 streams: 2190x real time vs 2753x with 4). Start the server with --batch_inference "
                            "(--batch_max_size 8..12, --batch_lanes 2..4) or add GPUs with --devices.")
        if vad_weights:
            _vad.configure(vad_weights, self.devices[0])         # silero_vad.onnx or .npz -> Silero on every GPU that serves a client
        return BackendType(backend)

    def run(self, host, port=9090, backend="hip", faster_whisper_custom_model_path=None, whisper_tensorrt_path=None,
            trt_multilingual=False, trt_py_session=False, single_model=False, max_clients=4, max_connection_time=600,
            cache_path="~/.cache/whisper-live/", rest_port=8000, enable_rest=False, cors_origins: Optional[str] = None,
            batch_enabled=False, batch_max_size=8, batch_window_ms=50, raw_pcm_input=False, metrics_port: int = 0,
            api_key: Optional[str] = None, rate_limit_rpm: int = 0, segment_post_processor=None,
            devices: Optional[Sequence[int]] = None, model_factory=None, ready: Optional[threading.Event] = None,
            vad_weights: Optional[str] = None, batch_lanes: int = 2):
        """Serve until ``shutdown()``. Same arguments as the reference (server.py:600-622) plus ``devices`` (GPU
        indices to shard connections over), ``model_factory`` and ``ready`` (set once the socket is listening;
        ``self.port`` then holds the bound port — pass ``port=0`` for an ephemeral one)."""
        backend_type = self.configure(backend, faster_whisper_custom_model_path, whisper_tensorrt_path, single_model,
                                      max_clients, max_connection_time, cache_path, enable_rest, batch_enabled,
                                      batch_max_size, batch_window_ms, raw_pcm_input, segment_post_processor, devices,
                                      model_factory, vad_weights, batch_lanes)
        if metrics_port > 0:
            wl_metrics.start_metrics_server(metrics_port)
        extra = {}
        if api_key:
            extra["process_request"] = functools.partial(_websocket_auth, api_key)
        handler = functools.partial(self.recv_audio, backend=backend_type,
                                    faster_whisper_custom_model_path=faster_whisper_custom_model_path,
                                    whisper_tensorrt_path=whisper_tensorrt_path, trt_multilingual=trt_multilingual,
                                    trt_py_session=trt_py_session)
        with ws.serve(handler, host, port, **extra) as server:
            self._server, self.port = server, server.port
            if ready is not None:
                ready.set()
            server.serve_forever()

    def shutdown(self):
        if self._server is not None:
            self._server.shutdown()
        for w in list(ServeClientHIP.BATCH_WORKERS.values()):
            w.stop()
        ServeClientHIP.BATCH_WORKERS.clear()


def main(argv=None):
    """``python -m whisperlive_amd.server`` — the flags of the reference's run_server.py that apply to this path."""
    import argparse
    ap = argparse.ArgumentParser(description="WhisperLive-protocol transcription server on MI355X")
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", "-p", type=int, default=9090)
    ap.add_argument("--backend", "-b", default="hip", choices=BackendType.valid_types())
    ap.add_argument("--model_path", "--faster_whisper_custom_model_path", "-fw", dest="model_path", default=None,
                    help="model directory (CTranslate2 model.bin or Hugging Face safetensors) + tokenizer.json")
    ap.add_argument("--no_single_model", "-nsm", action="store_true")
    ap.add_argument("--max_clients", type=int, default=4)
    ap.add_argument("--max_connection_time", type=int, default=600)
    ap.add_argument("--batch_inference", action="store_true")
    ap.add_argument("--batch_max_size", type=int, default=8)
    ap.add_argument("--batch_window_ms", type=int, default=50)
    ap.add_argument("--batch_lanes", type=int, default=2, help="worker lanes per GPU in --batch_inference mode (each lane: own engine slot and hardware "
                                                                "queue). Measured THROUGH the worker, 64 clips of 30 s, Whisper-large-v3 shapes, one MI355X, "
                                                                "--batch_max_size 8: 1436x real time on one lane, 2215x on four; --batch_max_size 12: 1473x / 2261x "
                                                                "(profiles/r4lv3rt_*, r4lv3b_*)")
    ap.add_argument("--raw_pcm_input", action="store_true")
    ap.add_argument("--metrics_port", type=int, default=0)
    ap.add_argument("--api_key", default=os.environ.get("WHISPERLIVE_API_KEY"))
    ap.add_argument("--devices", default="0", help="comma-separated GPU indices to shard connections over")
    ap.add_argument("--vad_weights", default=os.environ.get("WLX_SILERO_VAD_ONNX") or os.environ.get("WLX_SILERO_VAD_NPZ"),
                    help="Silero VAD weights: the silero_vad.onnx the reference downloads, or an .npz export of it "
                         "(python -m whisperlive_amd.silero_export); without it use_vad is refused with a WARNING")
    a = ap.parse_args(argv)
    TranscriptionServer().run(
        a.host, port=a.port, backend=a.backend, faster_whisper_custom_model_path=a.model_path,
        single_model=not a.no_single_model, max_clients=a.max_clients, max_connection_time=a.max_connection_time,
        batch_enabled=a.batch_inference, batch_max_size=a.batch_max_size, batch_window_ms=a.batch_window_ms, batch_lanes=a.batch_lanes,
        raw_pcm_input=a.raw_pcm_input, metrics_port=a.metrics_port, api_key=a.api_key,
        devices=[int(x) for x in a.devices.split(",") if x != ""], vad_weights=a.vad_weights)


if __name__ == "__main__":
    main()
